// count_mfma.hip — pair co-occurrence counts on the matrix cores (gfx950), for large dense blocks.
//
// The four numbers count.hip produces per column pair,
//     N = |Cx & Cy|   R = |Ax & Cy|   C = |Cx & Ay|   A = |Ax & Ay|,
// are the entries of the Gram matrix of {0,1} vectors: an int8 outer product over reads
// (BASELINE.json north_star: "MFMA only if the co-occurrence reduces to a dense int8 outer
// product" — it does).  v_mfma_i32_32x32x32_i8 accumulates in int32, so the counts stay exact.
//
// Weighted-bit operands.  Expanding bit planes into {0,1} bytes costs 3 VALU operations per 4 bytes (or an
// 8x larger image in LDS); the matrix cores do not need that.  For bit i of every byte of a raw plane dword W
//     x operand:  W & (0x01010101 << i)           = x_read * 2^i        in each byte
//     y operand:  Y & (0x01010101 << (6 - i))     = y_read * 2^(6 - i)  where Y = W with the bits of each byte
//                                                                        reversed, shifted down by one
// so every product is 64 * x * y and ONE v_and_b32 makes four operand bytes straight from the raw bits
// (i = 0..6; bit 7 takes a shift and a mask on the x side).  The accumulators hold 64 * count, exact below
// 2^26 reads per block (api.cpp keeps larger blocks on the VALU kernel).  No LDS, no barriers: each lane
// loads the raw (C, A) words of its own four columns from HBM/L2 and builds its own operands.
//
// One 256-thread workgroup (4 independent waves, 2 x 2) computes a 128 x 128 tile of a block's slot matrix;
// each wave owns 64 x 64 column pairs = 4 x-fragments (2 column groups x {C, A} plane) times 4 y-fragments,
// i.e. 16 accumulator tiles of 32 x 32 (256 accumulation registers).  Reads are taken 256 at a time (4 words):
// lanes 0..31 hold words 0, 1 and lanes 32..63 words 2, 3 of their column (the two k-halves of the MFMA), and
// the 8 bit positions give 8 MFMA k-steps of 32 reads, 16 MFMAs each.
//
// Lane maps (checked with exact integer data, tools/mfma_i8_probe.hip):
//   A operand: lane l holds A[row = l & 31][k = 16 (l >> 5) + j], j = 0..15 (bytes of 4 dwords)
//   B operand: lane l holds B[k = 16 (l >> 5) + j][col = l & 31]
//   C/D:       reg r of lane l is D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31]
// Which read a k index stands for is the same on both sides (byte j of dword q of half h, bit i: read
// 128 h + 32 q + 8 j + i of the 256), which is all the sum over k needs.
#include "mfma_common.h"

namespace lgmi {

typedef int v16i __attribute__((ext_vector_type(16)));

// bits of each byte reversed (bit i of byte b -> bit 7 - i of byte b)
__device__ __forceinline__ uint32_t rev_in_bytes(uint32_t w) { return __builtin_bswap32(__brev(w)); }

#define LGMI_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0)

__global__ __launch_bounds__(256, 1) void k_count_mfma(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint32_t* __restrict__ xlist, const uint32_t* __restrict__ ylist,
    const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    const ulonglong2* __restrict__ zero_entry, uint4* __restrict__ slots)
{
    const Tile t = tiles[xcd_remap_m(blockIdx.x, n_tiles)];
    const BlockPlan bp = plans[t.block];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wx = wave >> 1, wy = wave & 1u;        // 2 x 2 waves over the 128 x 128 tile
    const uint32_t lh = lane >> 5, r32 = lane & 31u;

    // the four columns this lane feeds: row r32 of x groups 0, 1 and of y groups 0, 1 of its wave
    MStageCol cx0, cx1, cy0, cy1;
    {
        const uint32_t rx = t.x0 + 64u * wx + r32, ry = t.y0 + 64u * wy + r32;
        cx0 = m_col(rx < bp.nx ? xlist[bp.xl_off + rx] : NONE, cols, cplanes);
        cx1 = m_col(rx + 32u < bp.nx ? xlist[bp.xl_off + rx + 32u] : NONE, cols, cplanes);
        cy0 = m_col(ry < bp.ny ? ylist[bp.yl_off + ry] : NONE, cols, cplanes);
        cy1 = m_col(ry + 32u < bp.ny ? ylist[bp.yl_off + ry + 32u] : NONE, cols, cplanes);
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

    // operand set of one k-step: x fragments (C, A of column group 0, C, A of group 1), y fragments likewise
    struct Ops { v4i a[4], b[4]; };
    // raw words of one 256-read step: x[c] / y[c] = plane quads (C0, A0, C1, A1) of this lane's 128-read half;
    // y is kept with the bits of every byte reversed
    struct Raw { v4i x[4], y[4]; };

#define LGMI_MFMA16(O)                                                                                \
    LGMI_MFMA(acc[0][0][0], O.a[0], O.b[0]); LGMI_MFMA(acc[0][0][1], O.a[1], O.b[0]);                   \
    LGMI_MFMA(acc[0][0][2], O.a[0], O.b[1]); LGMI_MFMA(acc[0][0][3], O.a[1], O.b[1]);                   \
    LGMI_MFMA(acc[0][1][0], O.a[0], O.b[2]); LGMI_MFMA(acc[0][1][1], O.a[1], O.b[2]);                   \
    LGMI_MFMA(acc[0][1][2], O.a[0], O.b[3]); LGMI_MFMA(acc[0][1][3], O.a[1], O.b[3]);                   \
    LGMI_MFMA(acc[1][0][0], O.a[2], O.b[0]); LGMI_MFMA(acc[1][0][1], O.a[3], O.b[0]);                   \
    LGMI_MFMA(acc[1][0][2], O.a[2], O.b[1]); LGMI_MFMA(acc[1][0][3], O.a[3], O.b[1]);                   \
    LGMI_MFMA(acc[1][1][0], O.a[2], O.b[2]); LGMI_MFMA(acc[1][1][1], O.a[3], O.b[2]);                   \
    LGMI_MFMA(acc[1][1][2], O.a[2], O.b[3]); LGMI_MFMA(acc[1][1][3], O.a[3], O.b[3]);
    // operands of bit I (0..6) of every byte: x * 2^I and y * 2^(6 - I)
#define LGMI_ANDS(O, X, YR, I)                                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        O.a[q_] = X[q_] & (0x01010101 << (I));                                                          \
        O.b[q_] = YR[q_] & (0x01010101 << (6 - (I)));                                                   \
    }
    // the two words of this lane's half of column C at step word KW -> plane quads (C, A)
#define LGMI_LOAD2(QC, QA, C, KW)                                                                     \
    {                                                                                                 \
        const uint4 e0_ = m_ld_entry(C, (KW), zero_entry), e1_ = m_ld_entry(C, (KW) + 1u, zero_entry);    \
        QC = v4i{(int)e0_.x, (int)e0_.y, (int)e1_.x, (int)e1_.y};                                       \
        QA = v4i{(int)e0_.z, (int)e0_.w, (int)e1_.z, (int)e1_.w};                                       \
    }
    // one slot = the 16 MFMAs of a k-step with V VALU operations of the next steps' preparation between them
    // (an MFMA holds vector issue for 8 of its 32 cycles); nothing moves across a slot boundary
#define LGMI_SLOT_END(V)                                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x002, (V), 0);                                            \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // One 256-read step: k-steps for bits 0..6 then bit 7 of the CUR words; meanwhile the NXT words are loaded
    // (slots 0, 1), bit-reversed (slot 5) and turned into the first operands of the next step (slot 7).
    // On entry P holds the operands of bit 0 of CUR and YR = (CUR.y >> 1) & 0x7F7F7F7F; on exit the same for NXT.
#define LGMI_STEP(CUR, NXT, KW)                                                                       \
    LGMI_LOAD2(NXT.x[0], NXT.x[1], cx0, (KW)) LGMI_LOAD2(NXT.x[2], NXT.x[3], cx1, (KW))                 \
    LGMI_ANDS(Q, CUR.x, yr, 1) LGMI_MFMA16(P) LGMI_SLOT_END(4)                                          \
    LGMI_LOAD2(NXT.y[0], NXT.y[1], cy0, (KW)) LGMI_LOAD2(NXT.y[2], NXT.y[3], cy1, (KW))                 \
    LGMI_ANDS(P, CUR.x, yr, 2) LGMI_MFMA16(Q) LGMI_SLOT_END(4)                                          \
    LGMI_ANDS(Q, CUR.x, yr, 3) LGMI_MFMA16(P) LGMI_SLOT_END(2)                                          \
    LGMI_ANDS(P, CUR.x, yr, 4) LGMI_MFMA16(Q) LGMI_SLOT_END(2)                                          \
    LGMI_ANDS(Q, CUR.x, yr, 5) LGMI_MFMA16(P) LGMI_SLOT_END(2)                                          \
    LGMI_ANDS(P, CUR.x, yr, 6) LGMI_MFMA16(Q)                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_)                                                    \
        NXT.y[q_] = v4i{(int)rev_in_bytes((uint32_t)NXT.y[q_].x), (int)rev_in_bytes((uint32_t)NXT.y[q_].y),  \
                        (int)rev_in_bytes((uint32_t)NXT.y[q_].z), (int)rev_in_bytes((uint32_t)NXT.y[q_].w)}; \
    LGMI_SLOT_END(4)                                                                                  \
    /* bit 7 of every byte: x * 64 (shifted down one place), y * 1 (bit 0 of the reversed form) */       \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        Q.a[q_] = (CUR.x[q_] >> 1) & 0x40404040;                                                        \
        Q.b[q_] = CUR.y[q_] & 0x01010101;                                                               \
    }                                                                                                 \
    LGMI_MFMA16(P) LGMI_SLOT_END(3)                                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) yr[q_] = (NXT.y[q_] >> 1) & 0x7F7F7F7F;            \
    LGMI_ANDS(P, NXT.x, yr, 0) LGMI_MFMA16(Q) LGMI_SLOT_END(4)

    const uint32_t n_words = t.k1 - t.k0;
    const uint32_t n_pair = (n_words + 7u) / 8u;          // steps go in pairs; words past k1 are outside every band -> zeros
    Raw ra, rb;
    Ops P, Q;
    v4i yr[4];
    uint32_t kw = t.k0 + 2u * lh;                         // first of this lane's two words of the step
    LGMI_LOAD2(ra.x[0], ra.x[1], cx0, kw) LGMI_LOAD2(ra.x[2], ra.x[3], cx1, kw)
    LGMI_LOAD2(ra.y[0], ra.y[1], cy0, kw) LGMI_LOAD2(ra.y[2], ra.y[3], cy1, kw)
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
        ra.y[q_] = v4i{(int)rev_in_bytes((uint32_t)ra.y[q_].x), (int)rev_in_bytes((uint32_t)ra.y[q_].y),
                       (int)rev_in_bytes((uint32_t)ra.y[q_].z), (int)rev_in_bytes((uint32_t)ra.y[q_].w)};
        yr[q_] = (ra.y[q_] >> 1) & 0x7F7F7F7F;
    }
    LGMI_ANDS(P, ra.x, yr, 0)
    for (uint32_t s = 0; s < n_pair; ++s) {
        LGMI_STEP(ra, rb, kw + 4u)
        LGMI_STEP(rb, ra, kw + 8u)
        kw += 8u;
    }

    // ---- epilogue: reg r of lane l is (x row (r&3) + 8 (r>>2) + 4 (l>>5), y col l&31) of its 32 x 32 tile;
    //      the accumulators hold 64 * count
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
                        slots[o] = make_uint4((uint32_t)acc[i][j][0][r] >> 6, (uint32_t)acc[i][j][1][r] >> 6,
                                              (uint32_t)acc[i][j][2][r] >> 6, (uint32_t)acc[i][j][3][r] >> 6);
                    }
                }
            }
        }
    }
}

void launch_count_mfma(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                       const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                       const ulonglong2* cplanes, const ulonglong2* zero_entry, uint4* slots)
{
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_count_mfma, dim3(n_tiles), dim3(256), 0, st, n_tiles, tiles, plans, xlist, ylist,
                       cols, cplanes, zero_entry, slots);
}

}  // namespace lgmi
