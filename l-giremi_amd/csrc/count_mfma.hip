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
// loaded from HBM as 16-byte (C, A) words by the thread that owns the column; that thread expands
// each 64-read word into 128 int8 {0,1} (3 VALU ops per 4 bytes: bfe, mul_u24 by 0x204081, and
// 0x01010101) and writes them to LDS in MFMA operand order, once per workgroup — every expanded
// fragment is consumed by the two waves that share the column group.  The 8x larger byte matrix
// exists only in LDS, two words deep.
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
static const int MCH = 6;      // 64-bit words a thread keeps in registers between global loads (multiple of 2 and 3)

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

#ifndef LGMI_ABL
#define LGMI_ABL 0      // timing-only ablations: 1 no expansion, 2 no per-word barrier, 4 no MFMA
#endif
#if LGMI_ABL & 4
#define LGMI_MFMA(acc, a, b) acc[0] += a.x + b.y
#else
#define LGMI_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0)
#endif

// LDS image of one word: [half h][fragment f][lane] of 16 bytes, where fragment f = side*8 + group*2 + plane
// (side 0 = x columns, 1 = y columns; group = 32-column group of the 128; plane 0 = C, 1 = A) and the entry
// of lane l = (k-half q = l >> 5, row r = l & 31) is exactly that lane's MFMA operand.
typedef uint4 ByteWord[2][16][64];   // 32 KB

__global__ __launch_bounds__(256, 1) void k_count_mfma(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint32_t* __restrict__ xlist, const uint32_t* __restrict__ ylist,
    const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    uint32_t* __restrict__ sN, uint32_t* __restrict__ sR, uint32_t* __restrict__ sC,
    uint32_t* __restrict__ sA)
{
    // three word buffers: while word w is multiplied, word w + 1 is already complete (its operands are
    // prefetched one half ahead, so no LDS wait ever blocks the single wave of a SIMD) and word w + 2 is written
    __shared__ ByteWord byt[3];

    const Tile t = tiles[xcd_remap_m(blockIdx.x, n_tiles)];
    const BlockPlan bp = plans[t.block];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wx = wave >> 1, wy = wave & 1u;        // 2 x 2 waves over the 128 x 128 tile

    // ---- producer role: thread tid owns column slot tid (0..127 x columns, 128..255 y columns)
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
    // where this thread's 8 entries of a word go: fragment pair (C, A) of its side and column group, row r
    const uint32_t pf = (tid >> 7) * 8u + ((tid & 127u) >> 5) * 2u, pr = tid & 31u;
    // ---- consumer role: fragments of this wave
    const uint32_t fa = (2u * wx) * 2u, fb = 8u + (2u * wy) * 2u;   // first x / y fragment (C plane of group 0)

    v16i acc[2][2][4];     // [x group][y group][N, R, C, A]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0;

#define LGMI_PRODUCE(BUF, E)                                                                          \
    {                                                                                                 \
        byt[BUF][0][pf][pr] = __builtin_bit_cast(uint4, expand16((E).x, 0u));                           \
        byt[BUF][0][pf][32u + pr] = __builtin_bit_cast(uint4, expand16((E).x, 16u));                    \
        byt[BUF][1][pf][pr] = __builtin_bit_cast(uint4, expand16((E).y, 0u));                           \
        byt[BUF][1][pf][32u + pr] = __builtin_bit_cast(uint4, expand16((E).y, 16u));                    \
        byt[BUF][0][pf + 1u][pr] = __builtin_bit_cast(uint4, expand16((E).z, 0u));                      \
        byt[BUF][0][pf + 1u][32u + pr] = __builtin_bit_cast(uint4, expand16((E).z, 16u));               \
        byt[BUF][1][pf + 1u][pr] = __builtin_bit_cast(uint4, expand16((E).w, 0u));                      \
        byt[BUF][1][pf + 1u][32u + pr] = __builtin_bit_cast(uint4, expand16((E).w, 16u));               \
    }
#define LGMI_FETCH(F, BUF, H)                                                                         \
    F##ac0 = __builtin_bit_cast(v4i, byt[BUF][H][fa][lane]);                                            \
    F##aa0 = __builtin_bit_cast(v4i, byt[BUF][H][fa + 1u][lane]);                                       \
    F##ac1 = __builtin_bit_cast(v4i, byt[BUF][H][fa + 2u][lane]);                                       \
    F##aa1 = __builtin_bit_cast(v4i, byt[BUF][H][fa + 3u][lane]);                                       \
    F##bc0 = __builtin_bit_cast(v4i, byt[BUF][H][fb][lane]);                                            \
    F##ba0 = __builtin_bit_cast(v4i, byt[BUF][H][fb + 1u][lane]);                                       \
    F##bc1 = __builtin_bit_cast(v4i, byt[BUF][H][fb + 2u][lane]);                                       \
    F##ba1 = __builtin_bit_cast(v4i, byt[BUF][H][fb + 3u][lane]);
#define LGMI_MFMA16(F)                                                                                \
    LGMI_MFMA(acc[0][0][0], F##ac0, F##bc0); LGMI_MFMA(acc[0][0][1], F##aa0, F##bc0);                   \
    LGMI_MFMA(acc[0][0][2], F##ac0, F##ba0); LGMI_MFMA(acc[0][0][3], F##aa0, F##ba0);                   \
    LGMI_MFMA(acc[0][1][0], F##ac0, F##bc1); LGMI_MFMA(acc[0][1][1], F##aa0, F##bc1);                   \
    LGMI_MFMA(acc[0][1][2], F##ac0, F##ba1); LGMI_MFMA(acc[0][1][3], F##aa0, F##ba1);                   \
    LGMI_MFMA(acc[1][0][0], F##ac1, F##bc0); LGMI_MFMA(acc[1][0][1], F##aa1, F##bc0);                   \
    LGMI_MFMA(acc[1][0][2], F##ac1, F##ba0); LGMI_MFMA(acc[1][0][3], F##aa1, F##ba0);                   \
    LGMI_MFMA(acc[1][1][0], F##ac1, F##bc1); LGMI_MFMA(acc[1][1][1], F##aa1, F##bc1);                   \
    LGMI_MFMA(acc[1][1][2], F##ac1, F##ba1); LGMI_MFMA(acc[1][1][3], F##aa1, F##ba1);

    // words are taken MCH at a time: while chunk c is expanded and multiplied, chunk c + 1 is in flight from HBM
    const uint32_t n_words = t.k1 - t.k0;
    const uint32_t n_chunk = (n_words + MCH - 1) / MCH;
    uint4 cur[MCH], nxt[MCH];
    v4i p_ac0, p_aa0, p_ac1, p_aa1, p_bc0, p_ba0, p_bc1, p_ba1;   // operands of reads 0..31 of the current word
    v4i q_ac0, q_aa0, q_ac1, q_aa1, q_bc0, q_ba0, q_bc1, q_ba1;   // operands of reads 32..63
#pragma unroll
    for (int k = 0; k < MCH; ++k) cur[k] = m_ld_entry(sc, t.k0 + k);
    LGMI_PRODUCE(0, cur[0])
    LGMI_PRODUCE(1, cur[1])
    __syncthreads();
    LGMI_FETCH(p_, 0, 0)
    for (uint32_t c = 0; c < n_chunk; ++c) {
        const uint32_t kb = t.k0 + (c + 1) * MCH;           // words past k1 are outside every band -> zeros
#pragma unroll
        for (int k = 0; k < MCH; ++k) nxt[k] = m_ld_entry(sc, kb + k);
#pragma unroll
        for (int k = 0; k < MCH; ++k) {
            // word w = c * MCH + k lives in byt[k % 3] (MCH is a multiple of 3); word w + 1 is complete,
            // word w + 2 is produced now into the buffer word w - 1 left at the last barrier
            LGMI_FETCH(q_, k % 3, 1)
#if !(LGMI_ABL & 1)
            if (k + 2 < MCH) { LGMI_PRODUCE((k + 2) % 3, cur[k + 2]) } else { LGMI_PRODUCE((k + 2) % 3, nxt[k + 2 - MCH]) }
#endif
            LGMI_MFMA16(p_)
            LGMI_FETCH(p_, (k + 1) % 3, 0)
            LGMI_MFMA16(q_)
            // spread the producer's expansion VALU between the 32 MFMAs of the word (an MFMA holds vector
            // issue for 8 of its 32 cycles).  Pinning the LDS reads / writes as well measured slower.
#pragma unroll
            for (int g_ = 0; g_ < 32; ++g_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // 5 VALU
            }
#if !(LGMI_ABL & 2)
            __syncthreads();
#endif
        }
#pragma unroll
        for (int k = 0; k < MCH; ++k) cur[k] = nxt[k];
    }

    // ---- epilogue: reg r of lane l is (x row (r&3) + 8 (r>>2) + 4 (l>>5), y col l&31) of its 32 x 32 tile
    const uint32_t lh = lane >> 5, r32 = lane & 31u;
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
