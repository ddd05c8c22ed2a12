// ecdf.hip — the reference's own p-value machinery: stat.ecdf (src/giremi/stat.py:7-29), used for
// the `mip` column at src/giremi/script/giremi.py:415-429:
//   f = ecdf(reference sample);  f(v) = #{sample < v} / len(sample)   (np.searchsorted side='left').
// Device radix sort of the sample (rocPRIM through hipCUB) + one binary search per query.
#include <hipcub/hipcub.hpp>

#include "lgmi_internal.h"

namespace lgmi {

__global__ void k_ecdf_eval(uint64_t n_query, const double* __restrict__ query, const double* __restrict__ sorted,
                            uint32_t m, double* __restrict__ out)
{
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_query) return;
    const double v = query[q];
    if (!(v == v)) { out[q] = v; return; }     // NaN in, NaN out
    uint32_t lo = 0, hi = m;                   // first index with sorted[idx] >= v
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sorted[mid] < v) lo = mid + 1; else hi = mid;
    }
    // the reference does not divide: y = concatenate([0], linspace(1/n, 1, n)) and returns y[idx] (stat.py:19,26-27).
    // numpy's linspace is start + arange(n) * step with step = (stop - start) / (n - 1), the last element set to stop:
    // the same three roundings here (no contraction: -ffp-contract=off), so the value is the reference's bit for bit
    double y;
    if (lo == 0u) y = 0.0;
    else if (lo == m) y = 1.0;
    else {
        const double start = 1.0 / (double)m;
        const double step = (1.0 - start) / (double)(m - 1u);
        y = (double)(lo - 1u) * step + start;
    }
    out[q] = y;
}

size_t ecdf_sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const double*)nullptr, (double*)nullptr, (int)n);
    return bytes;
}

hipError_t launch_ecdf(hipStream_t st, uint32_t n_ref, const double* ref, double* sorted, void* temp,
                       size_t temp_bytes, uint64_t n_query, const double* query, double* out)
{
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, ref, sorted, (int)n_ref, 0, 64, st);
    if (e != hipSuccess) return e;
    if (n_query)
        hipLaunchKernelGGL(k_ecdf_eval, dim3((uint32_t)((n_query + 255) / 256)), dim3(256), 0, st, n_query, query,
                           sorted, n_ref, out);
    return hipGetLastError();
}

}  // namespace lgmi
