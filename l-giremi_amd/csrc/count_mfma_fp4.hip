// count_mfma_fp4.hip — pair co-occurrence counts on the FP4 matrix cores (gfx950), blocks below 2^24 reads.
//
// Same Gram-matrix formulation and the same tiling as count_mfma.hip (int8), on
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands FP4 (e2m1) and unit block scales: K = 64 reads per
// instruction at twice the int8 rate (35.6 cycles per 32x32x64 measured, tools/mfma_fp4_probe.hip — 9 POP/s).
// The accumulators are f32: sums of {0,1} products are exact below 2^24, which api.cpp guarantees per block.
//
// Weighted-bit operands, nibble version.  An e2m1 nibble with exactly one of its low three bits set is a power
// of two — 0b0001 = 0.5 (the subnormal), 0b0010 = 1, 0b0100 = 2 — and 0b1000 is -0.  So for bit i of every
// nibble of a raw plane dword X (x side) / Y (y side) the operand dwords are
//     i = 0:  X & 0x11111111 (0.5)        (Y << 2) & 0x44444444 (2)
//     i = 1:  X & 0x22222222 (1)           Y & 0x22222222       (1)
//     i = 2:  X & 0x44444444 (2)          (Y >> 2) & 0x11111111 (0.5)
//     i = 3: (X >> 3) & 0x11111111 (0.5)  (Y >> 1) & 0x44444444 (2)
// every product is exactly x * y, and 12 VALU operations turn a raw dword pair into 8 operand dwords (32 reads
// on each side).  One operand = 4 dwords = 32 nibbles per lane; lanes 0..31 carry reads 0..127 of a 256-read
// step and lanes 32..63 reads 128..255, identically on both sides, which is all the sum over k needs.
// Lane maps and the format / scale codes are checked with exact data by tools/mfma_fp4_probe.hip.
#include "mfma_common.h"

namespace lgmi {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// cbsz = blgp = 4: FP4 e2m1 on both sides; scale bytes 0x7F = 2^0; only the first 4 dwords of an operand are read
#define LGMI_MFMA4(acc, a, b)                                                                          \
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0},            \
                                                          v8i{b.x, b.y, b.z, b.w, 0, 0, 0, 0}, acc, 4, 4, 0, \
                                                          0x7F7F7F7F, 0, 0x7F7F7F7F)

__global__ __launch_bounds__(256, 1) void k_count_mfma_fp4(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint32_t* __restrict__ xlist, const uint32_t* __restrict__ ylist,
    const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    const ulonglong2* __restrict__ zero_entry, uint32_t* __restrict__ sN, uint32_t* __restrict__ sR, uint32_t* __restrict__ sC,
    uint32_t* __restrict__ sA)
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

    v16f acc[2][2][4];     // [x group][y group][N, R, C, A]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][c][r] = 0.0f;

    // operand set of one k-step: x fragments (C, A of column group 0, C, A of group 1), y fragments likewise
    struct Ops { v4i a[4], b[4]; };
    // raw words of one 256-read step: x[c] / y[c] = plane quads (C0, A0, C1, A1) of this lane's 128-read half
    struct Raw { v4i x[4], y[4]; };

#define LGMI_MFMA16(O)                                                                                \
    LGMI_MFMA4(acc[0][0][0], O.a[0], O.b[0]); LGMI_MFMA4(acc[0][0][1], O.a[1], O.b[0]);                 \
    LGMI_MFMA4(acc[0][0][2], O.a[0], O.b[1]); LGMI_MFMA4(acc[0][0][3], O.a[1], O.b[1]);                 \
    LGMI_MFMA4(acc[0][1][0], O.a[0], O.b[2]); LGMI_MFMA4(acc[0][1][1], O.a[1], O.b[2]);                 \
    LGMI_MFMA4(acc[0][1][2], O.a[0], O.b[3]); LGMI_MFMA4(acc[0][1][3], O.a[1], O.b[3]);                 \
    LGMI_MFMA4(acc[1][0][0], O.a[2], O.b[0]); LGMI_MFMA4(acc[1][0][1], O.a[3], O.b[0]);                 \
    LGMI_MFMA4(acc[1][0][2], O.a[2], O.b[1]); LGMI_MFMA4(acc[1][0][3], O.a[3], O.b[1]);                 \
    LGMI_MFMA4(acc[1][1][0], O.a[2], O.b[2]); LGMI_MFMA4(acc[1][1][1], O.a[3], O.b[2]);                 \
    LGMI_MFMA4(acc[1][1][2], O.a[2], O.b[3]); LGMI_MFMA4(acc[1][1][3], O.a[3], O.b[3]);
#ifndef LGMI_ABL
#define LGMI_ABL 0      // timing-only ablations (tools/abl_mfma.sh): 1 operands never rebuilt, 2 no loads at all, 4 no loads inside the loop
#endif
    // operands of bit I (0..3) of every nibble (table in the header)
#if LGMI_ABL & 1
#define LGMI_OPS(O, R, I)
#else
#define LGMI_OPS(O, R, I)                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        if ((I) == 0) { O.a[q_] = R.x[q_] & 0x11111111; O.b[q_] = (R.y[q_] << 2) & 0x44444444; }         \
        if ((I) == 1) { O.a[q_] = R.x[q_] & 0x22222222; O.b[q_] = R.y[q_] & 0x22222222; }                \
        if ((I) == 2) { O.a[q_] = R.x[q_] & 0x44444444; O.b[q_] = (R.y[q_] >> 2) & 0x11111111; }         \
        if ((I) == 3) { O.a[q_] = (R.x[q_] >> 3) & 0x11111111; O.b[q_] = (R.y[q_] >> 1) & 0x44444444; }  \
    }
#endif
    // the two words of this lane's half of column C at step word KW -> plane quads (C, A)
#if LGMI_ABL & 2
#define LGMI_LOAD2(QC, QA, C, KW) { QC = v4i{(int)(KW), 1, 2, 3}; QA = v4i{4, 5, (int)(KW), 7}; }
#else
#define LGMI_LOAD2(QC, QA, C, KW)                                                                     \
    {                                                                                                 \
        const uint4 e0_ = m_ld_entry(C, (KW), zero_entry), e1_ = m_ld_entry(C, (KW) + 1u, zero_entry);    \
        QC = v4i{(int)e0_.x, (int)e0_.y, (int)e1_.x, (int)e1_.y};                                       \
        QA = v4i{(int)e0_.z, (int)e0_.w, (int)e1_.z, (int)e1_.w};                                       \
    }
#endif
    // one slot = the 16 MFMAs of a k-step with V VALU operations of the next k-step's preparation between them;
    // nothing moves across a slot boundary
#define LGMI_SLOT_END(V)                                                                              \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x002, (V), 0);                                            \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // One 256-read step of the CUR words (4 k-steps); the words two steps ahead are loaded into FAR during
    // slots 0 and 1, and slot 3 prepares bit 0 of the NXT words (loaded one step ago).
    // On entry P holds the operands of bit 0 of CUR; on exit those of NXT.
#if LGMI_ABL & 4
#define LGMI_LOADX(FAR, KW) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) FAR.x[q_] = FAR.x[q_] ^ (int)(KW);
#define LGMI_LOADY(FAR, KW) _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) FAR.y[q_] = FAR.y[q_] ^ (int)(KW);
#else
#define LGMI_LOADX(FAR, KW) LGMI_LOAD2(FAR.x[0], FAR.x[1], cx0, (KW)) LGMI_LOAD2(FAR.x[2], FAR.x[3], cx1, (KW))
#define LGMI_LOADY(FAR, KW) LGMI_LOAD2(FAR.y[0], FAR.y[1], cy0, (KW)) LGMI_LOAD2(FAR.y[2], FAR.y[3], cy1, (KW))
#endif
#define LGMI_STEP(CUR, NXT, FAR, KW)                                                                  \
    LGMI_LOADX(FAR, KW)                                                                               \
    LGMI_OPS(Q, CUR, 1) LGMI_MFMA16(P) LGMI_SLOT_END(4)                                                 \
    LGMI_LOADY(FAR, KW)                                                                               \
    LGMI_OPS(P, CUR, 2) LGMI_MFMA16(Q) LGMI_SLOT_END(5)                                                 \
    LGMI_OPS(Q, CUR, 3) LGMI_MFMA16(P) LGMI_SLOT_END(4)                                                 \
    LGMI_OPS(P, NXT, 0) LGMI_MFMA16(Q) LGMI_SLOT_END(3)

    const uint32_t n_words = t.k1 - t.k0;
    const uint32_t n_trip = (n_words + 11u) / 12u;        // steps go three at a time; words past k1 are outside every band -> zeros
    Raw ra, rb, rc;
#if LGMI_ABL & 4
    const uint32_t kw0_ = t.k0 + 2u * lh + 8u;
    LGMI_LOAD2(rc.x[0], rc.x[1], cx0, kw0_) LGMI_LOAD2(rc.x[2], rc.x[3], cx1, kw0_)
    LGMI_LOAD2(rc.y[0], rc.y[1], cy0, kw0_) LGMI_LOAD2(rc.y[2], rc.y[3], cy1, kw0_)
#endif
    Ops P, Q;
    uint32_t kw = t.k0 + 2u * lh;                         // first of this lane's two words of the step
    LGMI_LOAD2(ra.x[0], ra.x[1], cx0, kw) LGMI_LOAD2(ra.x[2], ra.x[3], cx1, kw)
    LGMI_LOAD2(ra.y[0], ra.y[1], cy0, kw) LGMI_LOAD2(ra.y[2], ra.y[3], cy1, kw)
    LGMI_LOAD2(rb.x[0], rb.x[1], cx0, kw + 4u) LGMI_LOAD2(rb.x[2], rb.x[3], cx1, kw + 4u)
    LGMI_LOAD2(rb.y[0], rb.y[1], cy0, kw + 4u) LGMI_LOAD2(rb.y[2], rb.y[3], cy1, kw + 4u)
    LGMI_OPS(P, ra, 0)
    for (uint32_t s = 0; s < n_trip; ++s) {
        LGMI_STEP(ra, rb, rc, kw + 8u)
        LGMI_STEP(rb, rc, ra, kw + 12u)
        LGMI_STEP(rc, ra, rb, kw + 16u)
        kw += 12u;
    }

    // ---- epilogue: reg r of lane l is (x row (r&3) + 8 (r>>2) + 4 (l>>5), y col l&31) of its 32 x 32 tile
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

void launch_count_mfma_fp4(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                           const uint32_t* xlist, const uint32_t* ylist, const Col* cols,
                           const ulonglong2* cplanes, const ulonglong2* zero_entry, uint32_t* sN, uint32_t* sR,
                           uint32_t* sC, uint32_t* sA)
{
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_count_mfma_fp4, dim3(n_tiles), dim3(256), 0, st, n_tiles, tiles, plans, xlist, ylist,
                       cols, cplanes, zero_entry, sN, sR, sC, sA);
}

}  // namespace lgmi
