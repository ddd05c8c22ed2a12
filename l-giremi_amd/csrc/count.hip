// count.hip — pair co-occurrence count kernel (gfx950).
//
// Replaces the per-pair dict rebuilding + set intersection + label building of
// the reference (src/giremi/mutual_information.py:15-40) and the contingency
// matrix inside sklearn.metrics.mutual_info_score (:41) by AND + popcount over
// packed bit planes: for columns x (rows of the slot matrix) and y,
//     N = |Cx & Cy|   R = |Ax & Cy|   C = |Cx & Ay|   A = |Ax & Ay|
// where C is the coverage plane and A the allele plane of a column.  The 3x3
// table of a site pair is assembled from these four numbers (and, for sites
// with class-0 reads, from the same numbers of their pseudo columns) in emit.hip.
//
// One 256-thread workgroup computes one 64 x 64 tile of a block's slot matrix,
// sweeping the tile's word range in LDS stages of KC words.  Each thread owns a
// 4 x 4 sub-tile (rows txl+16a, cols 4*tyl+b) and keeps its 64 counters in
// VGPRs.  VALU-bound: 16 pair-updates x 16 instructions (8 v_and_b32 +
// 8 v_bcnt_u32_b32 accumulate) per staged word per thread.
#include "lgmi_internal.h"

namespace lgmi {

// tile index remap: the dispatcher deals workgroups round-robin over the 8 XCDs
// (b % 8 labels the XCD group), so give each XCD a contiguous run of the tile
// list — neighbouring tiles share their x rows in that XCD's L2.  Bijective for
// any n (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    uint32_t q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct StageCol {  // what a thread needs to stage its 4 words of one column
    const ulonglong2* base;  // cplanes + off - w0  (so base[k] is word k)
    uint32_t w0, w1;         // band [w0, w1); empty when the slot is past the list
};

__device__ __forceinline__ uint4 ld_entry(const StageCol& c, uint32_t k) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k >= c.w0 && k < c.w1) v = *reinterpret_cast<const uint4*>(c.base + k);
    return v;
}

// v_bcnt_u32_b32 dst, src0, src1 computes popcount(src0) + src1: one instruction per 32-bit half.
// Written as inline asm because hipcc otherwise selects bcnt(a,0), bcnt(b,0) and a v_add3_u32
// for `acc += popc(a) + popc(b)` — 25 % more VALU instructions in a VALU-bound loop.
#define LGMI_CNT(acc, v) asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(v));
#define LGMI_PAIR(xe, ye, i)                                    \
    LGMI_CNT(accN[i], xe.x & ye.x) LGMI_CNT(accN[i], xe.y & ye.y) \
    LGMI_CNT(accR[i], xe.z & ye.x) LGMI_CNT(accR[i], xe.w & ye.y) \
    LGMI_CNT(accC[i], xe.x & ye.z) LGMI_CNT(accC[i], xe.y & ye.w) \
    LGMI_CNT(accA[i], xe.z & ye.z) LGMI_CNT(accA[i], xe.w & ye.w)

__global__ __launch_bounds__(256) void k_count(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint32_t* __restrict__ xlist, const uint32_t* __restrict__ ylist,
    const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    uint4* __restrict__ slots)
{
    // [buf][k][x rows 0..63 | y cols 64..127 (permuted)] of (C_lo, C_hi, A_lo, A_hi)
    __shared__ uint4 lds[2][KC][2 * TILE];

    const Tile t = tiles[xcd_remap(blockIdx.x, n_tiles)];
    const BlockPlan bp = plans[t.block];
    const uint32_t tid = threadIdx.x;
    const uint32_t tyl = tid & 15u;   // y group: cols 4*tyl .. 4*tyl+3
    const uint32_t txl = tid >> 4;    // x group: rows txl + 16a

    // ---- staging role: column slot cs (0..127), words kh*4 .. kh*4+3 of each stage
    const uint32_t cs = tid >> 1, kh = (tid & 1u) * 4u;
    StageCol sc;
    uint32_t lds_slot;
    {
        uint32_t col = NONE;
        if (cs < (uint32_t)TILE) {
            uint32_t r = t.x0 + cs;
            if (r < bp.nx) col = xlist[bp.xl_off + r];
            lds_slot = cs;
        } else {
            uint32_t c = cs - TILE, q = t.y0 + c;
            if (q < bp.ny) col = ylist[bp.yl_off + q];
            lds_slot = TILE + (c & 3u) * 16u + (c >> 2);  // thread's b-th col at b*16 + tyl
        }
        if (col != NONE) {
            Col ci = cols[col];
            sc.base = cplanes + ci.off - ci.w0;
            sc.w0 = ci.w0;
            sc.w1 = ci.w0 + ci.nw;
        } else {
            sc.base = cplanes;
            sc.w0 = 1u;
            sc.w1 = 0u;
        }
    }

    uint32_t accN[16], accR[16], accC[16], accA[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { accN[i] = 0; accR[i] = 0; accC[i] = 0; accA[i] = 0; }

    const uint32_t n_stage = (t.k1 - t.k0 + KC - 1) / KC;
    uint4 st0, st1, st2, st3;
    {
        uint32_t k = t.k0 + kh;
        st0 = ld_entry(sc, k); st1 = ld_entry(sc, k + 1); st2 = ld_entry(sc, k + 2); st3 = ld_entry(sc, k + 3);
        lds[0][kh + 0][lds_slot] = st0; lds[0][kh + 1][lds_slot] = st1;
        lds[0][kh + 2][lds_slot] = st2; lds[0][kh + 3][lds_slot] = st3;
    }
    __syncthreads();

    for (uint32_t s = 0; s < n_stage; ++s) {
        const uint32_t buf = s & 1u;
        const bool more = (s + 1 < n_stage);
        if (more) {  // next stage's global loads fly under this stage's popcounts
            uint32_t k = t.k0 + (s + 1) * KC + kh;
            st0 = ld_entry(sc, k); st1 = ld_entry(sc, k + 1); st2 = ld_entry(sc, k + 2); st3 = ld_entry(sc, k + 3);
        }
        // every stage is swept whole: words outside [k0, k1) lie beyond the union band of
        // the x columns or of the y columns, so one side is all zero and adds nothing.
#pragma unroll 1
        for (int k = 0; k < KC; ++k) {
            const uint4 x0 = lds[buf][k][txl], x1 = lds[buf][k][txl + 16];
            const uint4 x2 = lds[buf][k][txl + 32], x3 = lds[buf][k][txl + 48];
            const uint4 y0 = lds[buf][k][TILE + tyl], y1 = lds[buf][k][TILE + 16 + tyl];
            const uint4 y2 = lds[buf][k][TILE + 32 + tyl], y3 = lds[buf][k][TILE + 48 + tyl];
            LGMI_PAIR(x0, y0, 0)  LGMI_PAIR(x0, y1, 1)  LGMI_PAIR(x0, y2, 2)  LGMI_PAIR(x0, y3, 3)
            LGMI_PAIR(x1, y0, 4)  LGMI_PAIR(x1, y1, 5)  LGMI_PAIR(x1, y2, 6)  LGMI_PAIR(x1, y3, 7)
            LGMI_PAIR(x2, y0, 8)  LGMI_PAIR(x2, y1, 9)  LGMI_PAIR(x2, y2, 10) LGMI_PAIR(x2, y3, 11)
            LGMI_PAIR(x3, y0, 12) LGMI_PAIR(x3, y1, 13) LGMI_PAIR(x3, y2, 14) LGMI_PAIR(x3, y3, 15)
        }
        if (more) {
            lds[buf ^ 1u][kh + 0][lds_slot] = st0; lds[buf ^ 1u][kh + 1][lds_slot] = st1;
            lds[buf ^ 1u][kh + 2][lds_slot] = st2; lds[buf ^ 1u][kh + 3][lds_slot] = st3;
        }
        __syncthreads();
    }

    // ---- epilogue: 4 rows x (4 consecutive cols) per thread; a slot is (N, R, C, A), 16 bytes: 64 contiguous bytes per row
    const uint32_t col = t.y0 + 4u * tyl;
    if (col < bp.ny_pad) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const uint32_t row = t.x0 + txl + 16u * a;
            if (row < bp.nx) {
                const uint64_t o = bp.slot_base + (uint64_t)row * bp.ny_pad + col;
#pragma unroll
                for (int b = 0; b < 4; ++b) slots[o + b] = make_uint4(accN[4 * a + b], accR[4 * a + b], accC[4 * a + b], accA[4 * a + b]);
            }
        }
    }
}

void launch_count(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                  const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                  const ulonglong2* cplanes, uint4* slots)
{
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_count, dim3(n_tiles), dim3(256), 0, st, n_tiles, tiles, plans, xlist, ylist,
                       cols, cplanes, slots);
}

}  // namespace lgmi
