// count_mfma.hip — pair co-occurrence counts on the matrix cores (gfx950), for large dense blocks.
//
// The four numbers count.hip produces per column pair,
//     N = |Cx & Cy|   R = |Ax & Cy|   C = |Cx & Ay|   A = |Ax & Ay|,
// are the entries of the Gram matrix of {0,1} vectors: an int8 outer product over reads
// (BASELINE.json north_star: "MFMA only if the co-occurrence reduces to a dense int8 outer
// product" — it does).  v_mfma_i32_32x32x32_i8 accumulates in int32, so the counts stay exact.
//
// One 256-thread workgroup (4 waves, 2 x 2) computes a 128 x 128 tile of a block's slot matrix;
// each wave owns 64 x 64 column pairs = 4 x-fragments (2 column groups x {C, A} plane) times
// 4 y-fragments, i.e. 16 accumulator tiles of 32 x 32 (256 registers).  Bit planes are staged
// through LDS exactly as in count.hip (16-byte (C, A) entries); a lane expands the 16 bits of
// its row and k-half into 16 int8 {0,1} in registers (3 VALU ops per 4 bytes:
// bfe, mul_u24 by 0x204081, and 0x01010101) and feeds them straight to the MFMA — the 8x larger
// byte matrix never exists in HBM or LDS.
//
// Lane maps (checked with exact integer data, tools/mfma_i8_probe.hip):
//   A operand: lane l holds A[row = l & 31][k = 16 (l >> 5) + j], j = 0..15 (bytes of 4 dwords)
//   B operand: lane l holds B[k = 16 (l >> 5) + j][col = l & 31]
//   C/D:       reg r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
#include "lgmi_internal.h"

namespace lgmi {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static const int MT = 128;     // tile edge in columns
static const int MKC = 8;      // 64-bit words staged per LDS stage

__device__ __forceinline__ uint32_t xcd_remap_m(uint32_t b, uint32_t n) {
    uint32_t q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 16 bits (bit offset `sh` of word w) -> 16 bytes of 0/1: nibble n * 0x204081 puts bit i of the
// nibble at bit 8 i; no two partial products overlap, so there are no carries
__device__ __forceinline__ v4i expand16(uint32_t w, uint32_t sh) {
    v4i o;
    o.x = (int)(__umul24((w >> sh) & 0xFu, 0x204081u) & 0x01010101u);
    o.y = (int)(__umul24((w >> (sh + 4u)) & 0xFu, 0x204081u) & 0x01010101u);
    o.z = (int)(__umul24((w >> (sh + 8u)) & 0xFu, 0x204081u) & 0x01010101u);
    o.w = (int)(__umul24((w >> (sh + 12u)) & 0xFu, 0x204081u) & 0x01010101u);
    return o;
}

struct MStageCol { const ulonglong2* base; uint32_t w0, w1; };

__device__ __forceinline__ uint4 m_ld_entry(const MStageCol& c, uint32_t k) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k >= c.w0 && k < c.w1) v = *reinterpret_cast<const uint4*>(c.base + k);
    return v;
}

#define LGMI_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0)

__global__ __launch_bounds__(256, 1) void k_count_mfma(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint32_t* __restrict__ xlist, const uint32_t* __restrict__ ylist,
    const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    uint32_t* __restrict__ sN, uint32_t* __restrict__ sR, uint32_t* __restrict__ sC,
    uint32_t* __restrict__ sA)
{
    // [buf][k][slot]: slots 0..127 = x columns of the tile, 128..255 = y columns; (C_lo, C_hi, A_lo, A_hi)
    __shared__ uint4 lds[2][MKC][2 * MT];

    const Tile t = tiles[xcd_remap_m(blockIdx.x, n_tiles)];
    const BlockPlan bp = plans[t.block];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wx = wave >> 1, wy = wave & 1u;        // 2 x 2 waves over the 128 x 128 tile
    const uint32_t r32 = lane & 31u, q16 = (lane >> 5) * 16u;

    // ---- staging role: thread tid stages column slot tid, all MKC words of a stage
    MStageCol sc;
    {
        uint32_t col = NONE;
        if (tid < (uint32_t)MT) { const uint32_t r = t.x0 + tid; if (r < bp.nx) col = xlist[bp.xl_off + r]; }
        else { const uint32_t q = t.y0 + (tid - MT); if (q < bp.ny) col = ylist[bp.yl_off + q]; }
        if (col != NONE) {
            const Col ci = cols[col];
            sc.base = cplanes + ci.off - ci.w0; sc.w0 = ci.w0; sc.w1 = ci.w0 + ci.nw;
        } else { sc.base = cplanes; sc.w0 = 1u; sc.w1 = 0u; }
    }

    v16i acc[2][2][4];     // [x group][y group][N, R, C, A]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0;

    const uint32_t n_stage = (t.k1 - t.k0 + MKC - 1) / MKC;
    uint4 st[MKC];
#pragma unroll
    for (int k = 0; k < MKC; ++k) st[k] = m_ld_entry(sc, t.k0 + k);
#pragma unroll
    for (int k = 0; k < MKC; ++k) lds[0][k][tid] = st[k];
    __syncthreads();

    // Software pipeline inside the wave (one wave per SIMD, so nothing else hides the expansion):
    // while the 16 MFMAs of one 32-read half run on the matrix pipe, the VALU expands the fragments
    // of the next half (an MFMA holds vector issue for 8 of its 32 cycles: ~6 VALU slots per MFMA,
    // 96 per half = the 8 x 12 expansion instructions).  sched_group_barrier pins the 2 MFMA : 12 VALU
    // interleave that the source order suggests.
    const uint32_t xs0 = 64u * wx + r32, ys0 = MT + 64u * wy + r32;
#define LGMI_EXPAND8(F, X0, X1, Y0, Y1, LO)                                                   \
    F##ac0 = expand16(LO ? X0.x : X0.y, q16); F##aa0 = expand16(LO ? X0.z : X0.w, q16);        \
    F##ac1 = expand16(LO ? X1.x : X1.y, q16); F##aa1 = expand16(LO ? X1.z : X1.w, q16);        \
    F##bc0 = expand16(LO ? Y0.x : Y0.y, q16); F##ba0 = expand16(LO ? Y0.z : Y0.w, q16);        \
    F##bc1 = expand16(LO ? Y1.x : Y1.y, q16); F##ba1 = expand16(LO ? Y1.z : Y1.w, q16);
#define LGMI_MFMA16(F)                                                                         \
    LGMI_MFMA(acc[0][0][0], F##ac0, F##bc0); LGMI_MFMA(acc[0][0][1], F##aa0, F##bc0);          \
    LGMI_MFMA(acc[0][0][2], F##ac0, F##ba0); LGMI_MFMA(acc[0][0][3], F##aa0, F##ba0);          \
    LGMI_MFMA(acc[0][1][0], F##ac0, F##bc1); LGMI_MFMA(acc[0][1][1], F##aa0, F##bc1);          \
    LGMI_MFMA(acc[0][1][2], F##ac0, F##ba1); LGMI_MFMA(acc[0][1][3], F##aa0, F##ba1);          \
    LGMI_MFMA(acc[1][0][0], F##ac1, F##bc0); LGMI_MFMA(acc[1][0][1], F##aa1, F##bc0);          \
    LGMI_MFMA(acc[1][0][2], F##ac1, F##ba0); LGMI_MFMA(acc[1][0][3], F##aa1, F##ba0);          \
    LGMI_MFMA(acc[1][1][0], F##ac1, F##bc1); LGMI_MFMA(acc[1][1][1], F##aa1, F##bc1);          \
    LGMI_MFMA(acc[1][1][2], F##ac1, F##ba1); LGMI_MFMA(acc[1][1][3], F##aa1, F##ba1);
#define LGMI_INTERLEAVE()                                                                      \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   /* 2 MFMA  */                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);  /* 12 VALU */                     \
    }
    v4i p_ac0, p_aa0, p_ac1, p_aa1, p_bc0, p_ba0, p_bc1, p_ba1;   // fragments of the half in flight
    v4i n_ac0, n_aa0, n_ac1, n_aa1, n_bc0, n_ba0, n_bc1, n_ba1;   // fragments being expanded
    for (uint32_t s = 0; s < n_stage; ++s) {
        const uint32_t buf = s & 1u;
        const bool more = (s + 1 < n_stage);
        if (more) {
#pragma unroll
            for (int k = 0; k < MKC; ++k) st[k] = m_ld_entry(sc, t.k0 + (s + 1) * MKC + k);
        }
        uint4 x0 = lds[buf][0][xs0], x1 = lds[buf][0][xs0 + 32u];
        uint4 y0 = lds[buf][0][ys0], y1 = lds[buf][0][ys0 + 32u];
        LGMI_EXPAND8(p_, x0, x1, y0, y1, true)
#pragma unroll 1
        for (int k = 0; k < MKC; ++k) {
            const int kn = (k + 1 < MKC) ? k + 1 : k;       // the last word re-reads itself (result unused)
            const uint4 nx0 = lds[buf][kn][xs0], nx1 = lds[buf][kn][xs0 + 32u];
            const uint4 ny0 = lds[buf][kn][ys0], ny1 = lds[buf][kn][ys0 + 32u];
            // reads 0..31 of word k on the matrix pipe, reads 32..63 expanded meanwhile
            LGMI_EXPAND8(n_, x0, x1, y0, y1, false)
            LGMI_MFMA16(p_)
            LGMI_INTERLEAVE()
            // reads 32..63 of word k on the matrix pipe, reads 0..31 of word k + 1 expanded meanwhile
            LGMI_EXPAND8(p_, nx0, nx1, ny0, ny1, true)
            LGMI_MFMA16(n_)
            LGMI_INTERLEAVE()
            x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < MKC; ++k) lds[buf ^ 1u][k][tid] = st[k];
        }
        __syncthreads();
    }

    // ---- epilogue: reg r of lane l is (x row (r&3) + 8 (r>>2) + 4 (l>>5), y col l&31) of its 32 x 32 tile
    const uint32_t lh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t col = t.y0 + 64u * wy + 32u * j + r32;
            if (col < bp.ny_pad) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t row = t.x0 + 64u * wx + 32u * i + (r & 3) + 8 * (r >> 2) + 4u * lh;
                    if (row < bp.nx) {
                        const uint64_t o = bp.slot_base + (uint64_t)row * bp.ny_pad + col;
                        sN[o] = (uint32_t)acc[i][j][0][r];
                        sR[o] = (uint32_t)acc[i][j][1][r];
                        sC[o] = (uint32_t)acc[i][j][2][r];
                        sA[o] = (uint32_t)acc[i][j][3][r];
                    }
                }
            }
        }
    }
}

void launch_count_mfma(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                       const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                       const ulonglong2* cplanes, uint32_t* sN, uint32_t* sR, uint32_t* sC, uint32_t* sA)
{
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_count_mfma, dim3(n_tiles), dim3(256), 0, st, n_tiles, tiles, plans, xlist, ylist,
                       cols, cplanes, sN, sR, sC, sA);
}

}  // namespace lgmi
