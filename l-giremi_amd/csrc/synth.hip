// synth.hip — (1) layout prep: ABI lo/hi planes -> interleaved (coverage, allele)
// column pairs the count kernel streams; (2) device-side generator of the dense
// synthetic chromosome (lgmi_synth_spec, SURVEY §8d "dense" regime).
#include "lgmi_internal.h"
#include "philox.h"

namespace lgmi {

// ---------------------------------------------------------------- prep (uploaded batches)
__global__ __launch_bounds__(256) void k_tri_flags(uint32_t n_sites, const uint32_t* __restrict__ site_nw,
                                                   const uint64_t* __restrict__ site_plane_off,
                                                   const uint64_t* __restrict__ planes, uint8_t* __restrict__ tri)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_sites) return;
    const uint32_t nw = site_nw[s];
    const uint64_t* lo = planes + site_plane_off[s];
    const uint64_t* hi = lo + nw;
    uint64_t any = 0;
    for (uint32_t k = threadIdx.x; k < nw; k += blockDim.x) any |= lo[k] & hi[k];
    if (any) tri[s] = 1;  // benign race: every writer stores 1
}

void launch_tri_flags(hipStream_t st, uint32_t n_sites, const uint32_t* site_nw,
                      const uint64_t* site_plane_off, const uint64_t* planes, uint8_t* tri) {
    if (!n_sites) return;
    hipLaunchKernelGGL(k_tri_flags, dim3(n_sites), dim3(256), 0, st, n_sites, site_nw, site_plane_off, planes, tri);
}

__global__ __launch_bounds__(256) void k_prep_cols(uint32_t c0, uint32_t n_cols, uint32_t n_sites, const Col* __restrict__ cols,
                                                   const uint32_t* __restrict__ pseudo_site,
                                                   const uint64_t* __restrict__ site_plane_off,
                                                   const uint64_t* __restrict__ planes, ulonglong2* __restrict__ cplanes)
{
    if (blockIdx.x >= n_cols) return;
    const uint32_t c = c0 + blockIdx.x;              // columns [c0, c0 + n_cols): a pipelined upload preps them piece by piece
    const bool pseudo = c >= n_sites;
    const uint32_t s = pseudo ? pseudo_site[c - n_sites] : c;
    const Col ci = cols[c];
    const uint64_t* lo = planes + site_plane_off[s];
    const uint64_t* hi = lo + ci.nw;
    for (uint32_t k = threadIdx.x; k < ci.nw; k += blockDim.x) {
        const uint64_t l = lo[k], h = hi[k];
        ulonglong2 v;
        v.x = l | h;                       // covered
        v.y = pseudo ? (l & ~h) : (h & ~l); // class 1 (pseudo column) or class 2
        cplanes[ci.off + k] = v;
    }
}

void launch_prep_cols(hipStream_t st, uint32_t c0, uint32_t n_cols, uint32_t n_sites, const Col* cols,
                      const uint32_t* pseudo_site, const uint64_t* site_plane_off,
                      const uint64_t* planes, ulonglong2* cplanes) {
    if (!n_cols) return;
    hipLaunchKernelGGL(k_prep_cols, dim3(n_cols), dim3(256), 0, st, c0, n_cols, n_sites, cols, pseudo_site,
                       site_plane_off, planes, cplanes);
}

// *out = 1 when two flag arrays differ anywhere (a caller's lgmi_batch.site_tri against what the planes say)
__global__ void k_flags_differ(uint32_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int* __restrict__ out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && (a[k] != 0) != (b[k] != 0)) *out = 1;
}
void launch_flags_differ(hipStream_t st, uint32_t n, const uint8_t* a, const uint8_t* b, int* out) {
    if (n) hipLaunchKernelGGL(k_flags_differ, dim3((n + 255u) / 256u), dim3(256), 0, st, n, a, b, out);
}

// ---------------------------------------------------------------- dense synthetic chromosome
// raw allele of read r at site s: 0 ref, 1 alt, 2 third; one Philox call per cell
__device__ __forceinline__ void synth_word(const lgmi_synth_spec& sp, const SynthSite& ss, uint32_t s,
                                           uint32_t w, uint64_t& cov, uint64_t& a1, uint64_t& a2)
{
    cov = 0; a1 = 0; a2 = 0;
    const uint32_t k0 = (uint32_t)sp.seed, k1 = (uint32_t)(sp.seed >> 32);
    for (uint32_t b = 0; b < 64u; ++b) {
        const uint32_t r = w * 64u + b;
        if (r >= sp.n_reads) break;
        const U4 o = philox4x32_10(r, s, TAG_CELL, 0u, k0, k1);
        if ((o.x & 0xFFFFu) < sp.dropout_u16) continue;
        cov |= 1ull << b;
        uint32_t allele;
        if (ss.het) {
            const U4 h = philox4x32_10(r, 0xFFFFFFFFu, TAG_HAP, 0u, k0, k1);
            allele = (h.x & 1u) ^ (((o.y & 0xFFFFu) < sp.het_noise_u16) ? 1u : 0u);
        } else {
            allele = ((o.y & 0xFFFFu) < ss.e16) ? 1u : 0u;
        }
        if (ss.tri && (o.z & 0xFFFFu) < sp.tri_frac_u16) allele = 2u;
        if (allele == 1u) a1 |= 1ull << b;
        if (allele == 2u) a2 |= 1ull << b;
    }
}

__global__ __launch_bounds__(256) void k_synth_depth(lgmi_synth_spec sp, uint32_t W, uint32_t* __restrict__ depth3)
{
    const uint32_t chunks = (W + 255u) / 256u;
    const uint32_t s = blockIdx.x / chunks;
    const uint32_t w = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    const SynthSite ss = synth_site(sp, s);
    uint32_t d0 = 0, d1 = 0, d2 = 0;
    if (w < W) {
        uint64_t cov, a1, a2;
        synth_word(sp, ss, s, w, cov, a1, a2);
        d1 = __popcll(a1);
        d2 = __popcll(a2);
        d0 = __popcll(cov) - d1 - d2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        d0 += __shfl_xor(d0, o);
        d1 += __shfl_xor(d1, o);
        d2 += __shfl_xor(d2, o);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (d0) atomicAdd(&depth3[3 * s + 0], d0);
        if (d1) atomicAdd(&depth3[3 * s + 1], d1);
        if (d2) atomicAdd(&depth3[3 * s + 2], d2);
    }
}

void launch_synth_depth(hipStream_t st, const lgmi_synth_spec& sp, uint32_t W, uint32_t* depth3) {
    hipLaunchKernelGGL(k_synth_depth, dim3(((W + 255) / 256) * sp.n_sites), dim3(256), 0, st, sp, W, depth3);
}

// rank the alleles present in the site's depth dict (insertion order 0,1[,2]) by
// depth, descending, stable — mutual_information.py:25-32
__host__ __device__ inline void synth_rank(bool tri, const uint32_t d[3], uint32_t& major, uint32_t& minor) {
    const int n = tri ? 3 : 2;
    int best = 0;
    for (int a = 1; a < n; ++a) if (d[a] > d[best]) best = a;
    int second = -1;
    for (int a = 0; a < n; ++a) {
        if (a == best) continue;
        if (second < 0 || d[a] > d[second]) second = a;
    }
    major = (uint32_t)best;
    minor = (uint32_t)second;
}

__global__ __launch_bounds__(256) void k_synth_write(lgmi_synth_spec sp, uint32_t W, const uint32_t* __restrict__ depth3,
                                                     const uint32_t* __restrict__ pseudo_of_site,
                                                     ulonglong2* __restrict__ cplanes, uint32_t site_base)
{
    const uint32_t chunks = (W + 255u) / 256u;
    const uint32_t s = blockIdx.x / chunks;
    const uint32_t w = (blockIdx.x % chunks) * blockDim.x + threadIdx.x;
    if (w >= W) return;
    const SynthSite ss = synth_site(sp, s);
    uint64_t cov, a1, a2;
    synth_word(sp, ss, s, w, cov, a1, a2);
    const uint64_t a0 = cov & ~a1 & ~a2;
    const uint32_t d[3] = {depth3[3 * s], depth3[3 * s + 1], depth3[3 * s + 2]};
    uint32_t major, minor;
    synth_rank(ss.tri, d, major, minor);
    ulonglong2 v;
    v.x = cov;
    v.y = major == 0u ? a0 : (major == 1u ? a1 : a2);
    cplanes[(uint64_t)(site_base + s) * W + w] = v;     // depth3 / pseudo_of_site point at the block's first site
    const uint32_t pc = pseudo_of_site[s];
    if (pc != NONE) {
        v.y = minor == 0u ? a0 : (minor == 1u ? a1 : a2);
        cplanes[(uint64_t)pc * W + w] = v;
    }
}

void launch_synth_write(hipStream_t st, const lgmi_synth_spec& sp, uint32_t W, const uint32_t* depth3,
                        const uint32_t* pseudo_of_site, ulonglong2* cplanes, uint32_t site_base) {
    hipLaunchKernelGGL(k_synth_write, dim3(((W + 255) / 256) * sp.n_sites), dim3(256), 0, st, sp, W, depth3,
                       pseudo_of_site, cplanes, site_base);
}

}  // namespace lgmi
