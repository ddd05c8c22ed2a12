// mfma_common.h — column staging shared by the matrix-core count kernels (count_mfma.hip: int8,
// count_mfma_fp4.hip: FP4): tile -> XCD remap, the per-lane column cursor and its branch-free band load.
#pragma once
#include "lgmi_internal.h"

namespace lgmi {

typedef int v4i __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ uint32_t xcd_remap_m(uint32_t b, uint32_t n) {
    uint32_t q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct MStageCol { const ulonglong2* base; uint32_t w0, w1; };

static __device__ __forceinline__ MStageCol m_col(uint32_t col, const Col* __restrict__ cols, const ulonglong2* __restrict__ cplanes) {
    MStageCol sc;
    if (col != NONE) {
        const Col ci = cols[col];
        sc.base = cplanes + ci.off - ci.w0; sc.w0 = ci.w0; sc.w1 = ci.w0 + ci.nw;
    } else { sc.base = cplanes; sc.w0 = 1u; sc.w1 = 0u; }
    return sc;
}

// words outside a column's band read the all-zero entry api.cpp keeps after the last column: the load stays
// unconditional, so the loop has no branches and the compiler can count the loads in flight (s_waitcnt at
// first use, not right after the issue)
static __device__ __forceinline__ uint4 m_ld_entry(const MStageCol& c, uint32_t k, const ulonglong2* __restrict__ zero) {
    const ulonglong2* p = (k >= c.w0 && k < c.w1) ? c.base + k : zero;
    return *reinterpret_cast<const uint4*>(p);
}

}  // namespace lgmi
