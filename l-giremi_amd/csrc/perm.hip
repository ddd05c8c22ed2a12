// perm.hip — permutation-test p-value of every emitted site pair (gfx950).
//
// No counterpart in the reference (gxiaolab/L-GIREMI has no permutation test; its only
// p-value is the ECDF `mip`, src/giremi/stat.py:7-29).  BASELINE.json's north_star asks
// for it; the specification is DESIGN.md §5.
//
// Null: the class labels of site j are shuffled among the N common reads, i.e. the 3x3
// table is multivariate hypergeometric with the observed margins.  Statistic: MI,
// compared through S(T) = sum G[T_ab] with G[n] = round(n ln n * 2^28) in 64-bit integers
// (order-independent, so tables made of the same counts tie exactly).
// p = (1 + #{S(T_s) >= S(T_obs)}) / (n_shuffles + 1).
//
//   k_perm_fast     one lane per row.  Degenerate tables: p = 1.  2 x 2 tables (one
//                   degree of freedom k): shuffle s lands in the "as or more extreme" set
//                   iff u_s < P_tail, with P_tail summed exactly from log-factorials —
//                   pure ALU + Philox, no table is materialised.  Larger tables are
//                   queued for k_perm_general.
//   k_perm_general  one wave per queued row, shuffles spread over the 64 lanes; each
//                   shuffle draws the table with conditional hypergeometric draws (urn
//                   scheme for small samples, Stadlober's HRUA otherwise).
//
// Every outcome-deciding operation is integer arithmetic or IEEE +,-,*,/ on doubles
// (-ffp-contract=off) over host-built tables, so results do not depend on the device's
// libm; exp/log/sqrt are the explicit routines below.
#include "lgmi_internal.h"
#include "philox.h"

namespace lgmi {

static const uint32_t TAG_PERM2X2 = 0x5eed0004u;
static const uint32_t TAG_PERMGEN = 0x60000000u;

#define LN2_HI 6.93147180369123816490e-01
#define LN2_LO 1.90821492927058770002e-10
#define INV_LN2 1.44269504088896338700e+00

__device__ __forceinline__ double inv_n(int n) {   // 1/n, IEEE-rounded constants
    switch (n) {
        case 1: return 1.0;        case 2: return 1.0 / 2.0;   case 3: return 1.0 / 3.0;   case 4: return 1.0 / 4.0;
        case 5: return 1.0 / 5.0;  case 6: return 1.0 / 6.0;   case 7: return 1.0 / 7.0;   case 8: return 1.0 / 8.0;
        case 9: return 1.0 / 9.0;  case 10: return 1.0 / 10.0; case 11: return 1.0 / 11.0; case 12: return 1.0 / 12.0;
        case 13: return 1.0 / 13.0; case 14: return 1.0 / 14.0; case 15: return 1.0 / 15.0; case 17: return 1.0 / 17.0;
        case 19: return 1.0 / 19.0; case 21: return 1.0 / 21.0; case 23: return 1.0 / 23.0;
        default: return 0.0;
    }
}

__device__ __forceinline__ double det_exp(double x) {   // x <= 0
    if (!(x > -745.0)) return 0.0;
    if (x > 0.0) x = 0.0;
    const double k = floor(x * INV_LN2 + 0.5);
    const double r = (x - k * LN2_HI) - k * LN2_LO;
    double p = 1.0;
#pragma unroll
    for (int n = 11; n >= 1; --n) p = 1.0 + p * (r * inv_n(n));   // |r| <= 0.347: error < 1e-14
    const int ki = (int)k;
    if (ki >= -1000) return p * __longlong_as_double((long long)(ki + 1023) << 52);
    return (p * __longlong_as_double((long long)(-1000 + 1023) << 52)) *
           __longlong_as_double((long long)(ki + 1000 + 1023) << 52);
}

__device__ __forceinline__ double det_log(double x) {   // x > 0, normal
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = (m - 1.0) / (m + 1.0);
    const double f2 = f * f;
    double s = 0.0;
#pragma unroll
    for (int n = 23; n >= 1; n -= 2) s = s * f2 + inv_n(n);
    return (double)e * LN2_HI + (2.0 * f * s + (double)e * LN2_LO);
}

__device__ __forceinline__ double det_sqrt(double a) {   // a > 0: division-free Newton on 1/sqrt(a), then a*y
    const unsigned long long b = (unsigned long long)__double_as_longlong(a);
    double y = __longlong_as_double((long long)(0x5FE6EB50C7B537A9ull - (b >> 1)));
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        double t = a * y;
        t = t * y;
        t = 0.5 * t;
        t = 1.5 - t;
        y = y * t;
    }
    return a * y;
}

// ---------------------------------------------------------------- 2 x 2 exact tail
struct HG22 { uint32_t N, K, n, kmin, kmax; double c0; };

__device__ __forceinline__ long long stat22(const long long* __restrict__ G, const HG22& h, uint32_t k) {
    return G[h.N - h.K - h.n + k] + G[h.n - k] + G[h.K - k] + G[k];
}

__device__ __forceinline__ double pmf22(const double* __restrict__ LF, const HG22& h, uint32_t k) {
    double e = h.c0;
    e -= LF[k];
    e -= LF[h.K - k];
    e -= LF[h.n - k];
    e -= LF[h.N - h.K - h.n + k];
    return det_exp(e);
}

// lane-private part of the exact tail: bisection for the far boundary of the "as or more
// extreme" set {k <= klo} U {k >= khi}, and the choice between summing that set or its
// complement (whichever lies within ~7 sigma)
struct Tail22 { long long klo, khi; int centre; };

__device__ Tail22 bounds22(const long long* __restrict__ G, const HG22& h, uint32_t kobs) {
    const long long sobs = stat22(G, h, kobs);
    uint32_t kc = (uint32_t)(((unsigned long long)h.n * (unsigned long long)h.K) / (unsigned long long)h.N);
    if (kc < h.kmin) kc = h.kmin;
    if (kc > h.kmax) kc = h.kmax;
    Tail22 t;
    if (kobs <= kc) {
        long long lo = (long long)kc + 1, hi = (long long)h.kmax + 1;
        t.klo = kobs;
        while (lo < hi) {
            const long long mid = lo + (hi - lo) / 2;
            if (stat22(G, h, (uint32_t)mid) >= sobs) hi = mid; else lo = mid + 1;
        }
        t.khi = lo;
    } else {
        long long lo = (long long)h.kmin - 1, hi = (long long)kc;
        t.khi = kobs;
        while (lo < hi) {
            const long long mid = lo + (hi - lo + 1) / 2;
            if (stat22(G, h, (uint32_t)mid) >= sobs) lo = mid; else hi = mid - 1;
        }
        t.klo = lo;
    }
    const double var = (double)h.n * (double)h.K * (double)(h.N - h.K) * (double)(h.N - h.n) /
                       ((double)h.N * (double)h.N * (double)(h.N > 1 ? h.N - 1 : 1));
    const double clen = (double)(t.khi - t.klo - 1);
    t.centre = (clen * clen <= 49.0 * var + 64.0) ? 1 : 0;
    return t;
}

// wave-cooperative run: term number q goes to lane q % 64; after each 64-term chunk the
// run stops when its farthest term has fallen below 2^-44 of the run's first term.
// All control flow is wave-uniform (count, base and the two broadcast terms).
__device__ __forceinline__ void run_sum_wave(const double* __restrict__ LF, const HG22& h, long long k0, long long kend,
                                             int dir, int stop_rule, uint32_t lane, double& acc) {
    const long long count = dir > 0 ? kend - k0 + 1 : k0 - kend + 1;
    if (count <= 0) return;
    double first = 0.0;
    for (long long base = 0; base < count; base += 64) {
        const long long idx = base + lane;
        double term = 0.0;
        if (idx < count) {
            term = pmf22(LF, h, (uint32_t)(k0 + dir * idx));
            acc += term;
        }
        if (base == 0) first = __shfl(term, 0);
        if (stop_rule) {
            const long long rem = count - base - 1;
            const double last = __shfl(term, (int)(rem < 63 ? rem : 63));
            if (!(last >= first * 5.684341886080802e-14)) break;
        }
    }
}

__global__ __launch_bounds__(256) void k_perm_fast(
    uint64_t n_rows, const uint32_t* __restrict__ row_i, const uint32_t* __restrict__ row_j,
    const uint32_t* __restrict__ counts, const long long* __restrict__ G, const double* __restrict__ LF,
    uint32_t n_shuffles, uint64_t seed, double* __restrict__ out_p, uint32_t* __restrict__ out_exceed,
    uint32_t* __restrict__ gen_list, unsigned int* __restrict__ gen_count)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    // ---- phase A (one lane per row): classify, set up the hypergeometric, find the bounds
    int kind = 0;   // 0 nothing, 1 degenerate, 2 two-by-two, 3 queued for k_perm_general
    HG22 h = {1u, 0u, 0u, 0u, 0u, 0.0};
    Tail22 tb = {0, 0, 1};
    if (r < n_rows) {
        uint32_t T[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) T[k] = counts[9 * r + k];
        uint32_t R[3], C[3], N = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2];
            C[a] = T[a] + T[3 + a] + T[6 + a];
            N += R[a];
        }
        const int nr = (R[0] != 0) + (R[1] != 0) + (R[2] != 0), nc = (C[0] != 0) + (C[1] != 0) + (C[2] != 0);
        if (nr <= 1 || nc <= 1) {
            kind = 1;
        } else if (nr == 2 && nc == 2) {
            kind = 2;
            const int a2 = R[2] ? 2 : 1, b2 = C[2] ? 2 : 1;   // second non-empty row / column
            h.N = N; h.K = R[a2]; h.n = C[b2];
            h.kmin = h.K + h.n > N ? h.K + h.n - N : 0u;
            h.kmax = h.K < h.n ? h.K : h.n;
            h.c0 = LF[h.K];
            h.c0 += LF[N - h.K];
            h.c0 += LF[h.n];
            h.c0 += LF[N - h.n];
            h.c0 -= LF[N];
            tb = bounds22(G, h, T[3 * a2 + b2]);
        } else {
            kind = 3;
            gen_list[atomicAdd(gen_count, 1u)] = (uint32_t)r;
        }
    }
    // ---- phase B (the wave works on one row at a time): exact tail mass
    double my_p = 1.0;
    unsigned long long todo = __ballot(kind == 2);
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        HG22 hb;
        hb.N = __shfl(h.N, L); hb.K = __shfl(h.K, L); hb.n = __shfl(h.n, L);
        hb.kmin = __shfl(h.kmin, L); hb.kmax = __shfl(h.kmax, L); hb.c0 = __shfl(h.c0, L);
        const long long klo = __shfl(tb.klo, L), khi = __shfl(tb.khi, L);
        const int centre = __shfl(tb.centre, L);
        double acc = 0.0;
        if (centre) {
            run_sum_wave(LF, hb, klo + 1, khi - 1, +1, 0, lane, acc);
        } else {
            run_sum_wave(LF, hb, klo, (long long)hb.kmin, -1, 1, lane, acc);
            run_sum_wave(LF, hb, khi, (long long)hb.kmax, +1, 1, lane, acc);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        double p = centre ? 1.0 - acc : acc;
        if (p > 1.0) p = 1.0;
        if (p < 0.0) p = 0.0;
        if ((int)lane == L) my_p = p;
    }
    // ---- phase C (one lane per row): the shuffles
    if (kind == 0 || kind == 3) return;
    uint32_t exceed;
    if (kind == 1) {
        exceed = n_shuffles;
    } else {
        unsigned long long thr = (unsigned long long)(my_p * 4294967296.0);
        if (thr > 4294967296ull) thr = 4294967296ull;
        const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), ci = row_i[r], cj = row_j[r];
        exceed = 0;
        for (uint32_t s = 0; s < n_shuffles; s += 4u) {
            const U4 o = philox4x32_10(s >> 2, ci, cj, TAG_PERM2X2, k0, k1);
            exceed += ((unsigned long long)o.x < thr);
            if (s + 1u < n_shuffles) exceed += ((unsigned long long)o.y < thr);
            if (s + 2u < n_shuffles) exceed += ((unsigned long long)o.z < thr);
            if (s + 3u < n_shuffles) exceed += ((unsigned long long)o.w < thr);
        }
    }
    out_exceed[r] = exceed;
    out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
}

// ---------------------------------------------------------------- general tables
struct GenStream { uint32_t c0, c1, c2, k0, k1, call; U4 buf; int have; };

__device__ __forceinline__ double next_uniform(GenStream& g) {   // (0,1): (m + 0.5) * 2^-52
    if (g.have == 0) {
        g.buf = philox4x32_10(g.c0, g.c1, g.c2, TAG_PERMGEN + g.call, g.k0, g.k1);
        g.call++;
        g.have = 2;
    }
    const unsigned long long m = g.have == 2 ? (((unsigned long long)g.buf.x << 20) | (g.buf.y >> 12))
                                             : (((unsigned long long)g.buf.z << 20) | (g.buf.w >> 12));
    g.have--;
    return ((double)m + 0.5) * 2.220446049250313e-16;
}

#define HRUA_D1 1.7155277699214135
#define HRUA_D2 0.8989161620588988

// HRUA set-up of one (pop, good, sample); hoisted out of the shuffle loop for the one draw
// whose parameters do not depend on earlier draws (the values are what hg_draw computes)
struct HrSetup { uint32_t pop, good, sample; double d6, d8, d10, d11; };

__device__ __forceinline__ void hrua_setup(const double* __restrict__ LF, uint32_t pop, uint32_t good, uint32_t sample,
                                           HrSetup& hs) {
    const uint32_t bad = pop - good;
    const uint32_t m = sample < pop - sample ? sample : pop - sample;
    const uint32_t mn = good < bad ? good : bad, mx = good < bad ? bad : good;
    const double d4 = (double)mn / (double)pop, d5 = 1.0 - d4;
    hs.pop = pop; hs.good = good; hs.sample = sample;
    hs.d6 = (double)m * d4 + 0.5;
    const double d7 = det_sqrt((double)(pop - m) * (double)m * d4 * d5 / (double)(pop - 1u) + 0.5);
    hs.d8 = HRUA_D1 * d7 + HRUA_D2;
    const uint32_t d9 = (uint32_t)floor((double)(m + 1u) * (double)(mn + 1u) / ((double)pop + 2.0));
    hs.d10 = LF[d9] + LF[mn - d9] + LF[m - d9] + LF[mx - m + d9];
    const double cap = (double)((m < mn ? m : mn) + 1u);
    const double lim = floor(hs.d6 + 16.0 * d7);
    hs.d11 = cap < lim ? cap : lim;
}

__device__ __noinline__ uint32_t hg_draw(const double* __restrict__ LF, uint32_t pop, uint32_t good, uint32_t sample,
                                         GenStream& g, const HrSetup& cached) {
    const uint32_t bad = pop - good;
    const uint32_t m = sample < pop - sample ? sample : pop - sample;
    uint32_t z;
    if (sample == 0u || good == 0u) return 0u;
    if (bad == 0u) return sample;
    if (sample == pop) return good;
    if (m < 10u) {
        uint32_t rem_total = pop, rem_good = good, left = m;
        while (left > 0u && rem_good > 0u && rem_total > rem_good) {
            const double u = next_uniform(g);
            if ((uint32_t)(u * (double)rem_total) < rem_good) rem_good--;
            rem_total--;
            left--;
        }
        if (rem_total == rem_good) rem_good -= left;
        z = good - rem_good;
    } else {
        const uint32_t mn = good < bad ? good : bad, mx = good < bad ? bad : good;
        HrSetup hs;
        if (cached.pop == pop && cached.good == good && cached.sample == sample) hs = cached;
        else hrua_setup(LF, pop, good, sample, hs);
        for (;;) {
            const double x = next_uniform(g), y = next_uniform(g);
            const double w = hs.d6 + hs.d8 * (y - 0.5) / x;
            if (w < 0.0 || w >= hs.d11) continue;
            const uint32_t zc = (uint32_t)floor(w);
            const double tt = hs.d10 - (LF[zc] + LF[mn - zc] + LF[m - zc] + LF[mx - m + zc]);
            if (x * (4.0 - x) - 3.0 <= tt) { z = zc; break; }
            if (x * (x - tt) >= 1.0) continue;
            if (2.0 * det_log(x) <= tt) { z = zc; break; }
        }
        if (good > bad) z = m - z;
    }
    if (m < sample) z = good - z;
    return z;
}

__global__ __launch_bounds__(256) void k_perm_general(
    const uint32_t* __restrict__ gen_list, const unsigned int* __restrict__ gen_count,
    const uint32_t* __restrict__ row_i, const uint32_t* __restrict__ row_j, const uint32_t* __restrict__ counts,
    const long long* __restrict__ G, const double* __restrict__ LF, uint32_t n_shuffles, uint64_t seed,
    double* __restrict__ out_p, uint32_t* __restrict__ out_exceed)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t n_gen = *gen_count;
    for (uint32_t q = wave; q < n_gen; q += n_waves) {   // every wave reaches q >= n_gen: the grid drains
        const uint32_t r = gen_list[q];
        uint32_t T[9], R[3], C[3], N = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) T[k] = counts[9ull * r + k];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2];
            C[a] = T[a] + T[3 + a] + T[6 + a];
            N += R[a];
        }
        long long sobs = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) sobs += G[T[k]];
        // the first draw that is not trivially determined has the same parameters in every shuffle
        HrSetup fixed;
        fixed.pop = 0u; fixed.good = 0u; fixed.sample = 0u; fixed.d6 = 0.0; fixed.d8 = 0.0; fixed.d10 = 0.0; fixed.d11 = 0.0;
        {
            const uint32_t cb = C[0] ? C[0] : (C[1] ? C[1] : C[2]);          // first non-empty column
            uint32_t pop = N, good = R[0];
            if (good == 0u) { good = R[1]; }                                  // row 0 empty: its draw is trivial
            if (good == pop) { pop = 0u; }                                    // (cannot happen with >= 2 non-empty rows)
            const uint32_t m = cb < pop - cb ? cb : pop - cb;
            if (pop && good && good < pop && cb < pop && m >= 10u) hrua_setup(LF, pop, good, cb, fixed);
        }
        uint32_t exceed = 0;
        for (uint32_t s = lane; s < n_shuffles; s += 64u) {
            GenStream g;
            g.c0 = s; g.c1 = row_i[r]; g.c2 = row_j[r];
            g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.call = 0; g.have = 0;
            uint32_t rr0 = R[0], rr1 = R[1], pop_all = N;
            long long ss = 0;
#pragma unroll 1
            for (int b = 0; b < 3; ++b) {
                const uint32_t cb = b == 0 ? C[0] : (b == 1 ? C[1] : C[2]);
                uint32_t cc = cb, pop = pop_all;
                const uint32_t x0 = hg_draw(LF, pop, rr0, cc, g, fixed);
                pop -= rr0; cc -= x0;
                const uint32_t x1 = hg_draw(LF, pop, rr1, cc, g, fixed);
                cc -= x1;                        // the last row takes what is left of the column
                const uint32_t x2 = cc;
                rr0 -= x0; rr1 -= x1;
                pop_all -= cb;
                ss += G[x0] + G[x1] + G[x2];
            }
            exceed += (ss >= sobs);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) exceed += __shfl_xor(exceed, o);
        if (lane == 0) {
            out_exceed[r] = exceed;
            out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
        }
    }
}

void launch_perm(hipStream_t st, uint64_t n_rows, const uint32_t* out_i, const uint32_t* out_j,
                 const uint32_t* counts, const long long* G, const double* LF, uint32_t n_shuffles, uint64_t seed,
                 double* out_p, uint32_t* out_exceed, uint32_t* gen_list, unsigned int* gen_count)
{
    if (!n_rows) return;
    hipLaunchKernelGGL(k_perm_fast, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, n_rows, out_i, out_j,
                       counts, G, LF, n_shuffles, seed, out_p, out_exceed, gen_list, gen_count);
    // a fixed grid (8 blocks per CU) whose waves stride over the queued rows
    hipLaunchKernelGGL(k_perm_general, dim3(2048), dim3(256), 0, st, gen_list, gen_count, out_i, out_j, counts, G, LF,
                       n_shuffles, seed, out_p, out_exceed);
}

}  // namespace lgmi
