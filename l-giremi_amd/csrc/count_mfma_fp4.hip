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
//     i = 0:  X & 0x11111111 (0.5)         Y & 0x11111111       (0.5, block scale 2^2)
//     i = 1:  X & 0x22222222 (1)           Y & 0x22222222       (1)
//     i = 2:  X & 0x44444444 (2)           Y & 0x44444444       (2,   block scale 2^-2)
//     i = 3: (X >> 3) & 0x11111111 (0.5)  (Y >> 1) & 0x44444444 (2)
// every product is exactly x * y, and 10 VALU operations (12 before round 4: the y words of passes 0 and 2 were shifted
// into place instead of scaled) turn a raw dword pair into 8 operand dwords (32 reads
// on each side).  One operand = 4 dwords = 32 nibbles per lane; lanes 0..31 carry reads 0..127 of a 256-read
// step and lanes 32..63 reads 128..255, identically on both sides, which is all the sum over k needs.
// Lane maps and the format / scale codes are checked with exact data by tools/mfma_fp4_probe.hip.
#include "mfma_common.h"

namespace lgmi {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// cbsz = blgp = 4: FP4 e2m1 on both sides; only the first 4 dwords of an operand are read.  SB = the y side's four block
// scale bytes (E8M0: 0x7F = 2^0): round 4 lets the SCALE put a pass's product at 1 instead of a shift of the y words —
// bit i of a nibble is worth 2^(i-1) on either side, so x bit i times y bit i is 2^(2i-2) and the scale 2^(2-2i) makes it 1
// (i = 0: 0x81 = 2^2, i = 1: 0x7F, i = 2: 0x7D = 2^-2); the sign bit i = 3 still has to be moved on both sides.
#define LGMI_MFMA4S(acc, a, b, SB)                                                                     \
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v8i{a.x, a.y, a.z, a.w, 0, 0, 0, 0},            \
                                                          v8i{b.x, b.y, b.z, b.w, 0, 0, 0, 0}, acc, 4, 4, 0, \
                                                          0x7F7F7F7F, 0, (SB))

// ---- operand re-layout (once per run, ~2 ms at north-star): the column-major (C, A) entries of every 32-column
// group in the order the count kernel's waves load them, so that each of its loads is one contiguous kilobyte
// instead of 64 partial cache lines (the L1 tag pipeline was the count kernel's second bound).
// One workgroup = one group x two steps; thread -> (step, plane, lane).
__global__ __launch_bounds__(256) void k_gather_ops(
    const OpGroup* __restrict__ groups, const BlockPlan* __restrict__ plans, const uint32_t* __restrict__ xlist,
    const uint32_t* __restrict__ ylist, const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes,
    uint4* __restrict__ ops)
{
    const OpGroup g = groups[blockIdx.x];
    const BlockPlan bp = plans[g.block];
    const uint32_t step = 2u * blockIdx.y + (threadIdx.x >> 7);
    if (step >= bp.op_steps) return;
    const uint32_t plane = (threadIdx.x >> 6) & 1u, lane = threadIdx.x & 63u;
    const uint32_t idx = 32u * g.group + (lane & 31u);
    uint32_t col = NONE;
    if (g.is_y) { if (idx < bp.ny) col = ylist[bp.yl_off + idx]; }
    else        { if (idx < bp.nx) col = xlist[bp.xl_off + idx]; }
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    if (col != NONE) {
        const Col ci = cols[col];
        const uint32_t kw = 4u * step + 2u * (lane >> 5);
        const ulonglong2* base = cplanes + ci.off - ci.w0;
        uint4 e0 = make_uint4(0u, 0u, 0u, 0u), e1 = e0;
        if (kw >= ci.w0 && kw < ci.w0 + ci.nw) e0 = *reinterpret_cast<const uint4*>(base + kw);
        if (kw + 1u >= ci.w0 && kw + 1u < ci.w0 + ci.nw) e1 = *reinterpret_cast<const uint4*>(base + kw + 1u);
        out = plane ? make_uint4(e0.z, e0.w, e1.z, e1.w) : make_uint4(e0.x, e0.y, e1.x, e1.y);
    }
    const uint64_t o = (g.is_y ? bp.yop_off : bp.xop_off) + ((uint64_t)g.group * bp.op_steps + step) * 128u + 64u * plane + lane;
    ops[o] = out;
}

void launch_gather_ops(hipStream_t st, uint32_t n_groups, uint32_t max_steps, const OpGroup* groups, const BlockPlan* plans,
                       const uint32_t* xlist, const uint32_t* ylist, const Col* cols, const ulonglong2* cplanes,
                       uint4* ops)
{
    if (!n_groups || !max_steps) return;
    hipLaunchKernelGGL(k_gather_ops, dim3(n_groups, (max_steps + 1u) / 2u), dim3(256), 0, st, groups, plans, xlist, ylist,
                       cols, cplanes, ops);
}

__global__ __launch_bounds__(256, 1) void k_count_mfma_fp4(
    uint32_t n_tiles, const Tile* __restrict__ tiles, const BlockPlan* __restrict__ plans,
    const uint4* __restrict__ ops, uint4* __restrict__ slots)
{
    const Tile t = tiles[xcd_remap_m(blockIdx.x, n_tiles)];
    const BlockPlan bp = plans[t.block];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t wx = wave >> 1, wy = wave & 1u;        // 2 x 2 waves over the 128 x 128 tile
    const uint32_t lh = lane >> 5, r32 = lane & 31u;

    // this lane's stream of the four 32-column groups its wave feeds (x groups 0, 1 and y groups 0, 1): 16 bytes per
    // plane and step, the wave's 64 lanes contiguous (k_gather_ops); groups beyond the lists were never laid out and
    // are never stored from, so they read group 0 instead of unwritten memory
    const uint32_t t0 = t.k0 >> 2;                          // first 4-word step
    const uint32_t gxn = (bp.nx + 31u) / 32u, gyn = (bp.ny + 31u) / 32u;
    const v4i *cx0, *cx1, *cy0, *cy1;
    {
        const uint32_t gx = (t.x0 >> 5) + 2u * wx, gy = (t.y0 >> 5) + 2u * wy;
        // per-lane pointers; wave-uniform bases in scalar registers (saddr loads) were measured 2 % slower
        const v4i* xb = reinterpret_cast<const v4i*>(ops + bp.xop_off) + lane;
        const v4i* yb = reinterpret_cast<const v4i*>(ops + bp.yop_off) + lane;
        cx0 = xb + ((uint64_t)(gx < gxn ? gx : 0u) * bp.op_steps + t0) * 128u;
        cx1 = xb + ((uint64_t)(gx + 1u < gxn ? gx + 1u : 0u) * bp.op_steps + t0) * 128u;
        cy0 = yb + ((uint64_t)(gy < gyn ? gy : 0u) * bp.op_steps + t0) * 128u;
        cy1 = yb + ((uint64_t)(gy + 1u < gyn ? gy + 1u : 0u) * bp.op_steps + t0) * 128u;
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

#define LGMI_MFMA16(O, SB)                                                                            \
    LGMI_MFMA4S(acc[0][0][0], O.a[0], O.b[0], SB); LGMI_MFMA4S(acc[0][0][1], O.a[1], O.b[0], SB);       \
    LGMI_MFMA4S(acc[0][0][2], O.a[0], O.b[1], SB); LGMI_MFMA4S(acc[0][0][3], O.a[1], O.b[1], SB);       \
    LGMI_MFMA4S(acc[0][1][0], O.a[0], O.b[2], SB); LGMI_MFMA4S(acc[0][1][1], O.a[1], O.b[2], SB);       \
    LGMI_MFMA4S(acc[0][1][2], O.a[0], O.b[3], SB); LGMI_MFMA4S(acc[0][1][3], O.a[1], O.b[3], SB);       \
    LGMI_MFMA4S(acc[1][0][0], O.a[2], O.b[0], SB); LGMI_MFMA4S(acc[1][0][1], O.a[3], O.b[0], SB);       \
    LGMI_MFMA4S(acc[1][0][2], O.a[2], O.b[1], SB); LGMI_MFMA4S(acc[1][0][3], O.a[3], O.b[1], SB);       \
    LGMI_MFMA4S(acc[1][1][0], O.a[2], O.b[2], SB); LGMI_MFMA4S(acc[1][1][1], O.a[3], O.b[2], SB);       \
    LGMI_MFMA4S(acc[1][1][2], O.a[2], O.b[3], SB); LGMI_MFMA4S(acc[1][1][3], O.a[3], O.b[3], SB);
#ifndef LGMI_ABL
#define LGMI_ABL 0      // timing-only ablations (tools/abl_mfma.sh): 1 operands never rebuilt, 2 no loads at all, 4 no loads inside the loop,
                        // 8 loads always hit L1, 16 one load after every fourth MFMA, 32 non-temporal loads, 64 a barrier per trip
#endif
    // operands of bit I (0..3) of every nibble (table in the header)
#if LGMI_ABL & 1
#define LGMI_OPS(O, R, I)
#else
#define LGMI_OPS(O, R, I)                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                  \
        if ((I) == 0) { O.a[q_] = R.x[q_] & 0x11111111; O.b[q_] = R.y[q_] & 0x11111111; }                \
        if ((I) == 1) { O.a[q_] = R.x[q_] & 0x22222222; O.b[q_] = R.y[q_] & 0x22222222; }                \
        if ((I) == 2) { O.a[q_] = R.x[q_] & 0x44444444; O.b[q_] = R.y[q_] & 0x44444444; }                \
        if ((I) == 3) { O.a[q_] = (R.x[q_] >> 3) & 0x11111111; O.b[q_] = (R.y[q_] >> 1) & 0x44444444; }  \
    }
#endif
    // plane quads (C, A) of step ST (relative to the tile's first step) of group stream C
#if LGMI_ABL & 2
#define LGMI_LOAD2(QC, QA, C, ST) { QC = v4i{(int)(ST), 1, 2, 3}; QA = v4i{4, 5, (int)(ST), 7}; }
#elif LGMI_ABL & 8     // every load re-reads step 0: the same instructions, always L1 hits
#define LGMI_LOAD2(QC, QA, C, ST) { QC = C[0]; QA = C[64u]; }
#elif LGMI_ABL & 32    // streaming hint
#define LGMI_LOAD2(QC, QA, C, ST) { QC = __builtin_nontemporal_load(&C[(ST) * 128u]); QA = __builtin_nontemporal_load(&C[(ST) * 128u + 64u]); }
#else
#define LGMI_LOAD2(QC, QA, C, ST) { QC = C[(ST) * 128u]; QA = C[(ST) * 128u + 64u]; }
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
    // the same slot with one of its four loads after every fourth MFMA
#define LGMI_SLOT_END_L(V)                                                                            \
    _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
        __builtin_amdgcn_sched_group_barrier(0x002, (V), 0);                                            \
        if ((g_ & 3) == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
    }                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#if !(LGMI_ABL & 16)
#undef LGMI_SLOT_END_L
#define LGMI_SLOT_END_L(V) LGMI_SLOT_END(V)
#endif
// VALU operations the scheduler may place after each MFMA of a slot (the next pass's operands, address arithmetic)
#ifndef LGMI_V0
#define LGMI_V0 4
#define LGMI_V1 5
#define LGMI_V2 4
#define LGMI_V3 3
#endif
#define LGMI_STEP(CUR, NXT, FAR, KW)                                                                  \
    LGMI_LOADX(FAR, KW)                                                                               \
    LGMI_OPS(Q, CUR, 1) LGMI_MFMA16(P, 0x81818181) LGMI_SLOT_END_L(LGMI_V0)                             \
    LGMI_LOADY(FAR, KW)                                                                               \
    LGMI_OPS(P, CUR, 2) LGMI_MFMA16(Q, 0x7F7F7F7F) LGMI_SLOT_END_L(LGMI_V1)                             \
    LGMI_OPS(Q, CUR, 3) LGMI_MFMA16(P, 0x7D7D7D7D) LGMI_SLOT_END(LGMI_V2)                               \
    LGMI_OPS(P, NXT, 0) LGMI_MFMA16(Q, 0x7F7F7F7F) LGMI_SLOT_END(LGMI_V3)

    // Ring of five raw-word buffers: step s computes on ring[s % 5], prepares bit 0 of ring[(s + 1) % 5] and loads
    // step s + 4 into ring[(s + 4) % 5] — a load has ~3.7 steps (~3.5 us) to land.  With one wave per SIMD nothing else
    // hides memory latency: a three-buffer ring (1.7 steps) left the matrix pipe waiting on s_waitcnt (95 ms against
    // 81 ms for the same kernel without loads, tools/abl_mfma.sh).
    const uint32_t n_words = t.k1 - 4u * t0;
    const uint32_t n_trip = (n_words + 19u) / 20u;        // steps go five at a time; steps past the block's words hold zeros
    Raw ra, rb, rc, rd, re;
#if LGMI_ABL & 4
    LGMI_LOAD2(re.x[0], re.x[1], cx0, 4u) LGMI_LOAD2(re.x[2], re.x[3], cx1, 4u)
    LGMI_LOAD2(re.y[0], re.y[1], cy0, 4u) LGMI_LOAD2(re.y[2], re.y[3], cy1, 4u)
#endif
    Ops P, Q;
    uint32_t st = 0u;                                     // current step, relative to t0
#define LGMI_LOAD_ALL(R, ST)                                                                          \
    LGMI_LOAD2(R.x[0], R.x[1], cx0, (ST)) LGMI_LOAD2(R.x[2], R.x[3], cx1, (ST))                         \
    LGMI_LOAD2(R.y[0], R.y[1], cy0, (ST)) LGMI_LOAD2(R.y[2], R.y[3], cy1, (ST))
    // the barriers keep the prologue's loads in ring order: the loop's s_waitcnt vmcnt(n) counts loads in issue order,
    // and a prologue the compiler had regrouped by address made it wait for all but the newest 8 loads in every trip
    LGMI_LOAD_ALL(ra, 0u) __builtin_amdgcn_sched_barrier(0);
    LGMI_LOAD_ALL(rb, 1u) __builtin_amdgcn_sched_barrier(0);
    LGMI_LOAD_ALL(rc, 2u) __builtin_amdgcn_sched_barrier(0);
    LGMI_LOAD_ALL(rd, 3u) __builtin_amdgcn_sched_barrier(0);
    LGMI_OPS(P, ra, 0)
    for (uint32_t s = 0; s < n_trip; ++s) {
#if LGMI_ABL & 64
        __syncthreads();                                                   // the four waves re-aligned every five steps
#endif
#if LGMI_ABL & 8
        asm volatile("" : "+v"(cx0), "+v"(cx1), "+v"(cy0), "+v"(cy1));   // keeps the invariant loads inside the loop
#endif
        LGMI_STEP(ra, rb, re, st + 4u)
        LGMI_STEP(rb, rc, ra, st + 5u)
        LGMI_STEP(rc, rd, rb, st + 6u)
        LGMI_STEP(rd, re, rc, st + 7u)
        LGMI_STEP(re, ra, rd, st + 8u)
        st += 5u;
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
                        slots[o] = make_uint4((uint32_t)acc[i][j][0][r], (uint32_t)acc[i][j][1][r],
                                              (uint32_t)acc[i][j][2][r], (uint32_t)acc[i][j][3][r]);
                    }
                }
            }
        }
    }
}

void launch_count_mfma_fp4(hipStream_t st, uint32_t n_tiles, const Tile* tiles, const BlockPlan* plans,
                           const uint4* ops, uint4* slots)
{
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_count_mfma_fp4, dim3(n_tiles), dim3(256), 0, st, n_tiles, tiles, plans, ops, slots);
}

}  // namespace lgmi
