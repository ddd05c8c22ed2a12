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
//                   degree of freedom k): the mass P_tail of the "as or more extreme" set is
//                   summed exactly from log-factorials (units of 64 values dealt evenly over
//                   the wave, integer fixed-point sums), and the number of shuffles that land
//                   in the set is drawn as one Binomial(n_shuffles, P_tail) variate — pure
//                   ALU + Philox, no table is materialised.  Larger tables are queued for
//                   k_perm_general.
//   k_perm_general  a fixed grid of waves that take the queued rows from a shared counter, one
//                   wave per row, shuffles spread over the 64 lanes; each
//                   shuffle draws the table with conditional hypergeometric draws (urn
//                   scheme for small samples, Stadlober's HRUA otherwise).
//
// Every outcome-deciding operation is integer arithmetic or IEEE +,-,*,/ on doubles
// (-ffp-contract=off) over host-built tables; exp/log/sqrt are the explicit routines below.
// One shortcut: le_exp() lets the hardware's f32 exp decide the HRUA acceptances that are not
// within 1e-4 of the boundary (same decisions as det_exp as long as the hardware value is
// within 1e-4 of it — measured 2e-7, pinned by a device-side sweep in the GPU tests).
#include <algorithm>
#include <cstdlib>

#include "lgmi_internal.h"
#include "philox.h"

#ifndef LGMI_PABL
#define LGMI_PABL 0     // timing-only ablations of k_perm_fast (results wrong by construction), tools/abl_perm.sh.  Bits:
//   16 larger tables not queued (k_perm_fast alone)      32 exact-tail sums skipped      64 boundary search skipped
//  128 binomial draw skipped
// Every bit keeps all table indices inside the range the normal path uses.  (Rounds 1 - 3 had fifteen more meanings, for
// the sampling loops of k_perm_general: 1 2 4 8 1024 2048 4096 8192 16384 131072 262144, and 256 32768 65536, two of which
// hung or faulted.  Since round 4 those loops run ~5 % of the step — the six-cell rows take the perimeter walk — and the
// variants went; their measurements are in HISTORY.md.)
#endif
#define LGMI_PABL_KNOWN (16 | 32 | 64 | 128)
#if LGMI_PABL & ~LGMI_PABL_KNOWN
#error "LGMI_PABL: unknown ablation bit (16, 32, 64, 128 are left: see the list above)"
#endif

namespace lgmi {

// The two host-built tables, read as base (scalar registers) + 32-bit byte offset: the look-ups compile to
// `global_load_dwordx2 v, v_off, s[base]` — one shift per look-up instead of a 64-bit address in two more VGPRs.
// (Entries < 2^28: lgmi_run_device checks it.)
template <class T> struct Tab {
    const T* p;
    __device__ __forceinline__ T operator[](uint32_t i) const {
        return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + (size_t)(i * (uint32_t)sizeof(T)));
    }
    // the look-ups of the lock-step loop
    __device__ __forceinline__ T ls(uint32_t i) const {
        return (*this)[i];
    }
};
typedef Tab<long long> TabG;
typedef Tab<double> TabLF;

static const uint32_t TAG_PERM2X2 = 0x5eed0004u;
static const uint32_t TAG_PERMGEN = 0x60000000u;
static const uint32_t TAG_LSX = 0x70000000u;      // lock-step rows: the first draws X[k]
static const uint32_t TAG_LSC = 0x71000000u;      // lock-step rows: + lane = the lane's candidate stream

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

// exp(x) for x <= 0, relative error < 1e-12, made of fma, floor and exact scalings only
__device__ __forceinline__ double det_exp(double x) {
    if (!(x > -745.0)) return 0.0;
    if (x > 0.0) x = 0.0;
    const double k = floor(fma(x, INV_LN2, 0.5));
    double r = fma(-k, LN2_HI, x);
    r = fma(-k, LN2_LO, r);
    double p = 2.755731922398589e-07;
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 0.0001984126984126984);
    p = fma(p, r, 0.001388888888888889);
    p = fma(p, r, 0.008333333333333333);
    p = fma(p, r, 0.041666666666666664);
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ki = (int)k;
    if (ki >= -1000) return p * __longlong_as_double((long long)(ki + 1023) << 52);
    return (p * __longlong_as_double((long long)(-1000 + 1023) << 52)) *
           __longlong_as_double((long long)(ki + 1000 + 1023) << 52);
}

// x2 <= exp(t) with the decision of det_exp(t) at a fraction of its cost: the hardware's single-precision exp is
// within 1e-5 of det_exp (1 ulp of f32 plus the rounding of t, |t| < 87 where it is a normal number), so it
// decides every case that is not within 1e-4 of the boundary; only those go through det_exp.  When the f32 value
// underflows, exp(t) < 1.2e-38 is below any x2 the callers pass (x2 >= 2^-66).  The bound is not taken on trust:
// tests/test_gpu_parity.py::test_hardware_exp_stays_inside_the_guard sweeps t on the device (lgmi_selftest_le_exp)
// and checks |__expf / det_exp - 1| < 2.5e-5 and that le_exp never disagrees with det_exp on adversarial x2.
__device__ __forceinline__ bool le_exp(double x2, double t) {
    const double e = (double)__expf((float)t);
    bool r = x2 <= e * 0.9999;
    if (!r && x2 < e * 1.0001) r = x2 <= det_exp(t);    // the only branch: a handful of lanes per million
    return r;
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
    for (int n = 0; n < 4; ++n) {
        const double t = a * y;
        const double h = fma(-t, y, 1.0);
        y = fma(y * 0.5, h, y);
    }
    return a * y;
}

// ---------------------------------------------------------------- 2 x 2 exact tail
struct HG22 { uint32_t N, K, n, kmin, kmax; double c0; };

__device__ __forceinline__ long long stat22(TabG G, const HG22& h, uint32_t k) {
    return G[h.N - h.K - h.n + k] + G[h.n - k] + G[h.K - k] + G[k];
}

__device__ __forceinline__ double pmf22(TabLF LF, const HG22& h, uint32_t k) {
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
struct Tail22 { long long klo, khi; int centre; double var; };

// first k in [lo, hi] with pred(k), for a predicate that is false then true along k and true at hi (a sentinel: pred is
// never evaluated there).  Galloping from a guess g, then bisection inside the bracket: the result is the boundary of
// a monotone predicate, whatever the search order, so it is the one the specification's plain bisection finds — in
// ~6 probes instead of log2(range) ~ 16 when the guess is a few values off.  (Each probe is four dependent scattered
// look-ups of G[]: the plain bisection was 50 of k_perm_fast's 85 ms at north-star, tools/abl_perm.sh 80.)
// (32-bit integers: every count here is below 2^28)
template <class P>
__device__ __forceinline__ int first_true32(int lo, int hi, int g, P pred) {
    if (g < lo) g = lo;
    if (g > hi) g = hi;
    int l, r;                                          // pred(r) holds; pred(l) does not (or l == lo - 1)
    if (g == hi || pred(g)) {
        r = g; l = lo - 1;
        int step = 1;
        while (r > lo) {
            int c = r - step;
            if (c < lo) c = lo;
            if (pred(c)) { r = c; step <<= 1; } else { l = c; break; }
        }
    } else {
        l = g; r = hi;
        int step = 1;
        for (;;) {
            const int c = l + step;
            if (c >= hi) break;
            if (pred(c)) { r = c; break; }
            l = c; step <<= 1;
        }
    }
    while (r - l > 1) {
        const int mid = l + ((r - l) >> 1);
        if (pred(mid)) r = mid; else l = mid;
    }
    return r;
}

__device__ Tail22 bounds22(TabG G, const HG22& h, uint32_t kobs) {
    const long long sobs = stat22(G, h, kobs);
    uint32_t kc = (uint32_t)(((unsigned long long)h.n * (unsigned long long)h.K) / (unsigned long long)h.N);
    if (kc < h.kmin) kc = h.kmin;
    if (kc > h.kmax) kc = h.kmax;
    Tail22 t;
    // the statistic is (nearly) symmetric about its minimum n K / N: the mirror image of kobs is the guess
    const long long mirror = 2ll * (long long)kc + 1ll - (long long)kobs;
    if (kobs <= kc) {
        // first k in [kc + 1, kmax + 1] with S(k) >= sobs (kmax + 1: none)
        t.klo = kobs;
        t.khi = first_true32((int)kc + 1, (int)h.kmax + 1, (int)mirror,
                             [&](int k) { return stat22(G, h, (uint32_t)k) >= sobs; });
    } else {
        // last k in [kmin - 1, kc] with S(k) >= sobs (kmin - 1: none): the same search on the reflected axis
        const int lo = (int)h.kmin - 1, hi = (int)kc, refl = lo + hi;
        t.khi = kobs;
        t.klo = refl - first_true32(lo, hi, refl - (int)mirror, [&](int j) { return stat22(G, h, (uint32_t)(refl - j)) >= sobs; });
    }
    const double var = (double)h.n * (double)h.K * (double)(h.N - h.K) * (double)(h.N - h.n) /
                       ((double)h.N * (double)h.N * (double)(h.N > 1 ? h.N - 1 : 1));
    const double clen = (double)(t.khi - t.klo - 1);
    t.centre = (clen * clen <= 49.0 * var + 64.0) ? 1 : 0;
    t.var = var;
    return t;
}

// Exact mass of UNIT = 64 consecutive values k0 .. k0 + len - 1 in units of 2^-62: first term from the
// log-factorials (four table lines per unit — scattered 8-byte look-ups are what the L2 charges for), the
// following ones through the hypergeometric ratio carried division-free over sub-blocks of SUB = 16 steps
// (N <- N num, Q <- Q den, P <- fma(P, den, N); then rQ = 1 / Q, sum += (t P) rQ, t <- (t N) rQ), truncated to the fixed-point
// grid.  The mass of a range is the INTEGER sum of its units: it does not depend on which lane sums which unit
// or on the order of the additions (CPU specification: unit_mass / range_mass in oracle/lgmi_perm_oracle.c).
#ifndef LGMI_UNIT
#define LGMI_UNIT 64
#endif
static const uint32_t UNIT = LGMI_UNIT, SUB = 16;
static const uint32_t QBATCH = 4;     // rows a wave of k_perm_general takes from the shared counter at a time

__device__ __forceinline__ double unit_sum(TabLF LF, const HG22& h, uint32_t k0, uint32_t len) {   // any length (range_sum_d in the oracle)
    uint32_t k = k0;
    double term = pmf22(LF, h, k), sum = term;
    uint32_t rem = len - 1u;
#pragma unroll 1
    while (rem > 0u) {                                   // per-lane trip counts: lanes drop out, nothing is re-selected
        const uint32_t m = rem < SUB ? rem : SUB;
        double P = 0.0, Nn = 1.0, Q = 1.0;
        // num = (K-k)(n-k) and den = (k+1)(N-K-n+k+1) step by second differences: integers below 2^53, so the
        // sums are the same doubles as the products the specification writes
        const double a = (double)(h.K - k), b = (double)(h.n - k), c = (double)(k + 1u), d = (double)(h.N - h.K - h.n + k + 1u);
        double num = a * b, den = c * d, sn = a + b - 1.0, sd = c + d + 1.0;
        // (unrolling the full sub-blocks — no per-step loop test on the per-lane trip count, 7 instead of 9 vector
        //  instructions per value — was measured SLOWER, 56 ms against 51: the full and the partial sub-blocks of a
        //  wave's lanes then run one after the other; so were units of 128 or 256 values, 60 / 63 ms)
#ifndef LGMI_UNIT_UNROLL
#define LGMI_UNIT_UNROLL 2      // (1: 46.9 ms, 2: 46.1, 4: 46.6 for k_perm_fast at north-star; the same arithmetic in the same order)
#endif
#pragma unroll LGMI_UNIT_UNROLL
        for (uint32_t j = 0; j < m; ++j) {
            Nn = Nn * num;
            Q = Q * den;
            P = fma(P, den, Nn);
            num -= sn; den += sd; sn -= 2.0; sd += 2.0;
        }
        const double rQ = 1.0 / Q;                       // (round 4: one division per sub-block instead of two, both sides)
        sum += (term * P) * rQ;
        term = (term * Nn) * rQ;
        k += m;
        rem -= m;
    }
    return sum;
}
__device__ __forceinline__ unsigned long long unit_mass(TabLF LF, const HG22& h, uint32_t k0, uint32_t len) {
    return (unsigned long long)(unit_sum(LF, h, k0, len) * 4611686018427387904.0);   // 2^62
}

// The number of "as or more extreme" shuffles of a 2 x 2 table is Binomial(n_shuffles, P_tail): one binomial
// variate instead of n_shuffles Bernoulli trials (the CPU specification is binom_draw in
// oracle/lgmi_perm_oracle.c).  thr = trunc(P * 2^32).  n p < 10: sequential inversion (BINV), otherwise
// Hoermann's transformed rejection (BTRS) tested against the exact log-factorials.  One lane per row.
__device__ __forceinline__ uint32_t binom_draw(TabLF LF, uint32_t n, unsigned long long thr, uint32_t ci,
                               uint32_t cj, uint32_t k0, uint32_t k1) {
    if (thr == 0ull || n == 0u) return 0u;
    if (thr >= 4294967296ull) return n;
    const bool flip = thr > 2147483648ull;
    const uint32_t tt = flip ? (uint32_t)(4294967296ull - thr) : (uint32_t)thr;
    const double p = (double)tt * 2.3283064365386963e-10;
    const double q = 1.0 - p;
    const double np = (double)n * p;
    uint32_t trip = 0u, k = 0u;
    // trial t takes words (0, 1) of Philox call t / 2 when t is even, words (2, 3) of the same call when odd: the trials of a
    // wave's lanes run in step, so a call is made on every other trip only, and the first one serves both algorithms
    U4 o = philox4x32_10(0u, ci, cj, TAG_PERM2X2, k0, k1);
    if (np < 10.0) {
        const double qn = det_exp((double)n * det_log(q));
        const double lim = np + 10.0 * det_sqrt(np * q + 1.0);
        const uint32_t bound = lim < (double)n ? (uint32_t)lim : n;
        const double pq = p / q;
        for (;;) {
            const uint32_t w0 = (trip & 1u) ? o.z : o.x, w1 = (trip & 1u) ? o.w : o.y;
            // inversion on the scale of x!: U = x! (u - F(x - 1)), T = x! f(x) — no division per step
            double U = ((double)(((unsigned long long)w0 << 20) | (unsigned long long)(w1 >> 12)) + 0.5) * 2.220446049250313e-16;
            double T = qn;
            uint32_t x = 0u;
            while (U > T && x <= bound) {
                ++x;
                U = (U - T) * (double)x;
                T = T * ((double)(n - x + 1u) * pq);
            }
            if (x <= bound) { k = x; break; }
            if (!(++trip & 1u)) o = philox4x32_10(trip >> 1, ci, cj, TAG_PERM2X2, k0, k1);
        }
    } else {
        const double spq = det_sqrt(np * q);
        const double b = 1.15 + 2.53 * spq;
        const double a = -0.0873 + 0.0248 * b + 0.01 * p;
        const double c = np + 0.5;
        const double rb = 1.0 / b;                          // (round 4: one reciprocal for the two quotients by b, one per
        const double vr = 0.92 - 4.2 * rb;                  //  candidate for the two by us — five divisions per row were a
        const double alpha = (2.83 + 5.1 * rb) * spq;       //  quarter of this routine's instructions)
        const uint32_t m = (uint32_t)floor((double)(n + 1u) * p);
        const double lr = det_log(p / q);
        const double hm = LF[m] + LF[n - m];
        for (;;) {
            const uint32_t w0 = (trip & 1u) ? o.z : o.x, w1 = (trip & 1u) ? o.w : o.y;
            const double u = ((double)w0 + 0.5) * 2.3283064365386963e-10 - 0.5;
            double v = ((double)w1 + 0.5) * 2.3283064365386963e-10;
            const double us = 0.5 - fabs(u);
            const double rus = 1.0 / us;
            const double kf = floor((2.0 * a * rus + b) * u + c);
            if (kf >= 0.0 && kf <= (double)n) {
                k = (uint32_t)kf;
                if (us >= 0.07 && v <= vr) break;
                v = v * alpha / (a * rus * rus + b);
                const double h = hm - LF[k] - LF[n - k] + ((double)k - (double)m) * lr;
                if (le_exp(v, h)) break;                        // (the decision of v <= det_exp(h): v > 1e-29 here, see le_exp)
            }
            if (!(++trip & 1u)) o = philox4x32_10(trip >> 1, ci, cj, TAG_PERM2X2, k0, k1);
        }
    }
    return flip ? n - k : k;
}

// the value of the lane N places down the same 16-lane row (lanes whose source would lie outside the row keep their own)
template <int N> __device__ __forceinline__ double dpp_row_shr_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x110 + N, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x110 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// wave-uniform lane index -> v_readlane_b32 (no LDS round trip, unlike __shfl)
__device__ __forceinline__ uint32_t bcast32(uint32_t v, int L) { return (uint32_t)__builtin_amdgcn_readlane((int)v, L); }
__device__ __forceinline__ unsigned long long bcast64(unsigned long long v, int L) {
    return ((unsigned long long)bcast32((uint32_t)(v >> 32), L) << 32) | bcast32((uint32_t)v, L);
}

// (no amdgpu_waves_per_eu by default: pinned to 4 waves per SIMD the kernel took 53.7 ms, to 5 51.1, left to the compiler 50.1;
//  after the compaction of round 4 — 105 VGPRs — 47.0 as it is, 48.3 at five waves (96 + 2 spilled), 53.9 at six:
//  profiles/r04_perm_occupancy.txt.  -DLGMI_FAST_WPS=n pins it for such experiments.)
#ifdef LGMI_FAST_WPS
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LGMI_FAST_WPS, LGMI_FAST_WPS))) void k_perm_fast(PermArgs pa)
#else
__global__ __launch_bounds__(64) void k_perm_fast(PermArgs pa)
#endif
{
    const uint64_t n_rows = *pa.n_rows_dev;
    const uint32_t* __restrict__ row_i = pa.row_i; const uint32_t* __restrict__ row_j = pa.row_j;
    const TabG G{pa.G}; const TabLF LF{pa.LF};
    const uint32_t n_shuffles = pa.n_shuffles; const uint64_t seed = pa.seed;
    double* __restrict__ out_p = pa.out_p; uint32_t* __restrict__ out_exceed = pa.out_exceed;
    uint32_t* __restrict__ gen_list = pa.gen_list; unsigned int* __restrict__ gen_count = pa.gen_count;
    const uint32_t lane = threadIdx.x & 63u;
    // persistent one-wave workgroups: a fixed grid strides over the 64-row chunks (the row count is only known on
    // the device, and 7 million one-wave workgroups per launch were a cost of their own)
    const uint64_t n_chunks = (n_rows + 63ull) / 64ull;
    __shared__ uint32_t s_queue[128];                       // rows waiting to be queued for k_perm_general
    uint32_t qn = 0u;                                       // how many (wave-uniform)
    bool any_small = false;                                 // a queued row small enough for k_perm_enum (one store per wave, at the end)
    // Round 4: the rows that have tail sums to make are COMPACTED across chunks before the sums and the
    // binomial draw — 12 - 22 % of a chunk's lanes (larger tables, degenerate tables, linked pairs whose tail is below 2^-33,
    // centre forms with nothing between the bounds) used to sit idle through both phases.  Such rows are finished where they
    // are classified; the others wait in LDS (WREC dwords each) and are processed 64 at a time with every lane busy.
    enum { WREC = 12 };                                     // r, N, K, n, start1, len1, start2, len2, c0 (2), centre, -
    __shared__ __attribute__((aligned(16))) uint32_t s_work[128 * WREC];
    uint32_t wn = 0u;                                       // rows waiting (wave-uniform)
    __shared__ uint32_t s_pre[65];
    __shared__ unsigned long long s_acc[64];

    // phases B and C for the first `cnt` waiting rows (one lane per row)
    auto process = [&](uint32_t cnt) {
        const bool have = lane < cnt;
        uint32_t r = 0u, len1 = 0u, len2 = 0u;
        int centre = 1;
        if (have) {
            const uint32_t* w = s_work + lane * WREC;
            r = w[0]; len1 = w[5]; len2 = w[7];
            centre = (int)w[10];
        }
        // ---- phase B: exact mass of the "as or more extreme" set, or of its complement when that is the short side.
        //      Each row lists up to two ranges of k; the ranges are cut into units of 64 values and the units of the
        //      wave's rows are dealt to the lanes 64 at a time, so every lane has the same amount of work whatever
        //      the rows' range lengths are.  Unit masses are integers (2^-62) added with LDS atomics: exact, any order.
        const uint32_t units1 = (len1 + UNIT - 1u) / UNIT;
        const uint32_t my_units = units1 + (len2 + UNIT - 1u) / UNIT;
        uint32_t incl = my_units;                               // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += v;
        }
        s_pre[lane] = incl - my_units;
        if (lane == 63u) s_pre[64] = incl;
        s_acc[lane] = 0ull;
        __syncthreads();
        const uint32_t total = s_pre[64];                    // wave-uniform
        for (uint32_t base = 0; base < total; base += 64u) {
            const uint32_t id = base + lane;
            const bool active = id < total;
            // the row of unit id: the largest r with pre[r] <= id (its successor starts beyond id, so it owns units)
            uint32_t lo = 0u, hi = 63u;
            if (active) {
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    if (s_pre[mid] <= id) lo = mid; else hi = mid - 1u;
                }
            }
            const int rr = (int)lo;
            // the row's record straight from the list in LDS (three wide reads; ten ds_bpermute before)
            const uint4 wa = *reinterpret_cast<const uint4*>(s_work + (uint32_t)rr * WREC);
            const uint4 wb = *reinterpret_cast<const uint4*>(s_work + (uint32_t)rr * WREC + 4);
            const uint2 wc = *reinterpret_cast<const uint2*>(s_work + (uint32_t)rr * WREC + 8);
            HG22 hb;
            hb.N = wa.y; hb.K = wa.z; hb.n = wa.w;
            hb.kmin = 0u; hb.kmax = 0u;
            hb.c0 = __longlong_as_double((long long)(((unsigned long long)wc.y << 32) | wc.x));
            const uint32_t r_start1 = wb.x, r_len1 = wb.y, r_start2 = wb.z, r_len2 = wb.w;
            if (active) {
                uint32_t u = id - s_pre[rr];
                const uint32_t r_units1 = (r_len1 + UNIT - 1u) / UNIT;
                uint32_t k0, len;
                if (u < r_units1) { k0 = r_start1 + UNIT * u; len = r_len1 - UNIT * u; }
                else { u -= r_units1; k0 = r_start2 + UNIT * u; len = r_len2 - UNIT * u; }
                if (len > UNIT) len = UNIT;
                atomicAdd(&s_acc[rr], unit_mass(LF, hb, k0, len));
            }
        }
        __syncthreads();
        if (have) {
            const unsigned long long sm = s_acc[lane];
            if (pa.exact_2x2) {
                // the exact p: the mass itself (oracle/lgmi_perm_oracle.c: ptail22's p_out), no Monte-Carlo draw
                double p = centre ? 1.0 - (double)sm * 2.168404344971009e-19 : (double)sm * 2.168404344971009e-19;
                if (p > 1.0) p = 1.0;
                if (p < 0.0) p = 0.0;
                out_exceed[r] = LGMI_EXCEED_EXACT;
                out_p[r] = p;                                   // (the exact p is never derivable: out_p is there)
            } else {
                unsigned long long thr;
                if (centre) thr = sm <= 4611686018427387904ull ? (4611686018427387904ull - sm) >> 30 : 0ull;
                else { thr = sm >> 30; if (thr > 4294967296ull) thr = 4294967296ull; }
#if LGMI_PABL & 32
                thr = 1589137899ull;
#endif
                // ---- phase C: the shuffles
#if LGMI_PABL & 128
                const uint32_t exceed = (uint32_t)(thr >> 24);
#else
                const uint32_t exceed = binom_draw(LF, n_shuffles, thr, row_i[r] + pa.site_base, row_j[r] + pa.site_base, (uint32_t)seed, (uint32_t)(seed >> 32));
#endif
                out_exceed[r] = exceed;
                if (out_p) out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
            }
        }
        __syncthreads();                                        // s_pre / s_acc / s_work are reused
    };

    // (a fixed stride: taking the chunks from a shared counter, as k_perm_general takes its rows, was measured
    // slower here — 79 ms against 63 with 4 chunks per atomic, 139 ms with one)
    for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {   // wave-uniform trip count: the grid drains
    const uint64_t r = chunk * 64ull + lane;
    // ---- phase A (one lane per row): classify, set up the hypergeometric, find the bounds
    int kind = 0;   // 0 nothing, 1 degenerate, 2 two-by-two, 3 queued for k_perm_general
    HG22 h = {1u, 0u, 0u, 0u, 0u, 0.0};
    Tail22 tb = {0, 0, 1, 0.0};
    if (r < n_rows) {
        // the row's record from k_emit<2>: (N, K, n, kind << 30 | k_obs) — classification and margins were made where
        // the table was in registers (emit.hip: perm_record)
        const uint4 rc = pa.rec[r];
        kind = (int)(rc.w >> 30);
        if (kind == 2) {
            const uint32_t N = rc.x;
            h.N = N; h.K = rc.y; h.n = rc.z;
            h.kmin = h.K + h.n > N ? h.K + h.n - N : 0u;
            h.kmax = h.K < h.n ? h.K : h.n;
            h.c0 = LF[h.K];
            h.c0 += LF[N - h.K];
            h.c0 += LF[h.n];
            h.c0 += LF[N - h.n];
            h.c0 -= LF[N];
            const uint32_t kobs = rc.w & 0x3FFFFFFFu;
#if LGMI_PABL & 64
            tb.klo = kobs; tb.khi = tb.klo + 40 <= (long long)h.kmax + 1 ? tb.klo + 40 : (long long)h.kmax + 1; tb.centre = 1;   // stays inside the support
#else
            tb = bounds22(G, h, kobs);
#endif
        } else if (kind == 3) {
            // is any queued row small enough to be enumerated?  (k_perm_enum does nothing otherwise: at north-star its pass
            // over 1.7e7 queued rows, none of which qualifies, was 3 ms)
            if (rc.y <= pa.enum_max && (pa.exact_2x2 || (unsigned long long)rc.y <= 4ull * (unsigned long long)n_shuffles)) any_small = true;
#if !(LGMI_PABL & 16)
            if (!n_shuffles) { out_exceed[r] = LGMI_EXCEED_EXACT; if (out_p) out_p[r] = __longlong_as_double(0x7ff8000000000000ll); }   // exact_2x2 only: no estimate
#endif
        }
    }
#if !(LGMI_PABL & 16)
    if (n_shuffles || pa.exact_2x2) {                        // (exact mode: the exact paths for larger tables run without shuffles too)
        // queue the larger tables for k_perm_general.  They are collected in LDS over the wave's chunks and go out 64
        // at a time: one atomic on the queue counter per 64 queued rows.  (One atomic per chunk — 73 % of the chunks
        // have such a row — was 2.6 million atomics on one address, which the L2 serves at ~60 M/s: 14 of this
        // kernel's 63 ms, tools/abl_perm.sh 16.)  The order inside the queue is free.
        const unsigned long long qb = __ballot(kind == 3);
        if (qb) {
            if (kind == 3) s_queue[qn + (uint32_t)__popcll(qb & ((1ull << lane) - 1ull))] = (uint32_t)r;
            qn += (uint32_t)__popcll(qb);
            if (qn >= 64u) {
                __syncthreads();                                // (one wave per workgroup: a wave barrier)
                uint32_t base = 0u;
                if (lane == 0u) base = atomicAdd(gen_count, 64u);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                gen_list[base + lane] = s_queue[lane];
                const uint32_t rest = qn - 64u;                 // < 64: what stays for the next round
                const uint32_t keep = lane < rest ? s_queue[64u + lane] : 0u;
                __syncthreads();
                if (lane < rest) s_queue[lane] = keep;
                qn = rest;
                __syncthreads();
            }
        }
    }
#endif
    // ---- the ranges to sum; rows without any are finished here
    uint32_t start1 = 0, len1 = 0, start2 = 0, len2 = 0;
    if (kind == 2) {
#if !(LGMI_PABL & 32)
        if (tb.centre) {
            start1 = (uint32_t)(tb.klo + 1);
            len1 = (uint32_t)(tb.khi - tb.klo - 1);
        } else {
            // both tails start >= ~3.5 sigma out.  First (largest) terms below 2^-60: the whole set weighs less
            // than 2^-33, thr = 0.  Otherwise 8 sigma + 16 values per tail (the rest is below 2^-80 of it).
            const double f_lo = tb.klo >= (long long)h.kmin ? pmf22(LF, h, (uint32_t)tb.klo) : 0.0;
            const double f_hi = tb.khi <= (long long)h.kmax ? pmf22(LF, h, (uint32_t)tb.khi) : 0.0;
            if (!(f_lo < 8.673617379884035e-19 && f_hi < 8.673617379884035e-19)) {
                const long long D = (long long)(8.0 * det_sqrt(tb.var + 1.0)) + 16;
                long long lo_start = tb.klo - D + 1, hi_end = tb.khi + D - 1;
                if (lo_start < (long long)h.kmin) lo_start = (long long)h.kmin;
                if (hi_end > (long long)h.kmax) hi_end = (long long)h.kmax;
                if (tb.klo >= lo_start) { start1 = (uint32_t)lo_start; len1 = (uint32_t)(tb.klo - lo_start + 1); }
                if (hi_end >= tb.khi) { start2 = (uint32_t)tb.khi; len2 = (uint32_t)(hi_end - tb.khi + 1); }
            }
        }
#endif
    }
    const bool work = kind == 2 && (len1 | len2) != 0u;
    if ((kind == 1 || kind == 2) && !work) {
        // degenerate table: every shuffle gives it again.  2 x 2 with nothing to sum: the mass is 0 — the centre form's
        // complement is everything (thr = 2^32: every shuffle), the tail form's set weighs nothing (thr = 0: none).
        const bool all = kind == 1 || tb.centre;
        if (pa.exact_2x2) { out_exceed[r] = LGMI_EXCEED_EXACT; out_p[r] = all ? 1.0 : 0.0; }
        else {
#if LGMI_PABL & 32
            const uint32_t exceed = kind == 1 ? n_shuffles : (uint32_t)(1589137899ull >> 24);
#else
            const uint32_t exceed = all ? n_shuffles : 0u;
#endif
            out_exceed[r] = exceed;
            if (out_p) out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
        }
    }
    {
        const unsigned long long wb = __ballot(work);
        if (wb) {
            if (work) {
                uint32_t* w = s_work + (wn + (uint32_t)__popcll(wb & ((1ull << lane) - 1ull))) * WREC;
                const unsigned long long cb = (unsigned long long)__double_as_longlong(h.c0);
                w[0] = (uint32_t)r; w[1] = h.N; w[2] = h.K; w[3] = h.n; w[4] = start1; w[5] = len1; w[6] = start2; w[7] = len2;
                w[8] = (uint32_t)cb; w[9] = (uint32_t)(cb >> 32); w[10] = (uint32_t)tb.centre;
            }
            wn += (uint32_t)__popcll(wb);
            __syncthreads();
            if (wn >= 64u) {
                process(64u);
                const uint32_t rest = wn - 64u;                  // < 64: moved to the front
                uint32_t keep[WREC];
                if (lane < rest) {
#pragma unroll
                    for (int k = 0; k < WREC; ++k) keep[k] = s_work[(64u + lane) * WREC + k];
                }
                __syncthreads();
                if (lane < rest) {
#pragma unroll
                    for (int k = 0; k < WREC; ++k) s_work[lane * WREC + k] = keep[k];
                }
                wn = rest;
                __syncthreads();
            }
        }
    }
    }   // chunk loop
    if (wn) process(wn);
    // (thousands of waves storing to one address are served one after the other: 0.3 ms on the footprint batch — a wave
    //  stores only while it still reads 0 there)
    if (__any(any_small) && lane == 0u && __atomic_load_n(pa.gen_count + 3, __ATOMIC_RELAXED) == 0u) pa.gen_count[3] = 1u;
#if !(LGMI_PABL & 16)
    if (qn) {                                               // what is left of the wave's queue
        __syncthreads();
        uint32_t base = 0u;
        if (lane == 0u) base = atomicAdd(gen_count, qn);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (lane < qn) gen_list[base + lane] = s_queue[lane];
    }
#endif
}

// ---------------------------------------------------------------- small tables by enumeration
// The exact mass of the tables with S >= S_obs by ENUMERATION, then one binomial variate like a 2 x 2 row (specification:
// enum_plan / enum_mass in oracle/lgmi_perm_oracle.c).  The largest row and the largest column hold the dependent cells;
// the other four cells (a, b) run over 0 .. min(R[a], C[b]); a row is enumerated when the product of the four ranges is at
// most enum_max and 4 n_shuffles.  Rows of a few hundred reads with a rare third allele — what real footprints look like
// — are a few hundred tables (median 90 in the footprint batch of bench.py) at ~150 instructions each, against ~45 trips
// of k_perm_general's state machine for 1000 shuffles.
// A kernel of its own in front of k_perm_general (inside it the same code cost the sampling loops 3 % through register
// pressure).  The first ENUM_ROWS lanes read one queued row each and decide; the qualifying rows are then enumerated FOUR AT A TIME,
// sixteen lanes each (a wave per row was bound by the row's chain of dependent look-ups, and most rows do not fill 64
// lanes twice).  The rows that stay with k_perm_general go into a second list behind the first.
#ifndef LGMI_ENUM_ROWS
#define LGMI_ENUM_ROWS 16
#endif
static const uint32_t ENUM_ROWS = LGMI_ENUM_ROWS;   // <= 64
__global__ __launch_bounds__(64) void k_perm_enum(PermArgs pa)
{
    uint32_t* __restrict__ gen_list = pa.gen_list;
    const uint32_t* __restrict__ row_i = pa.row_i; const uint32_t* __restrict__ row_j = pa.row_j;
    const uint32_t* __restrict__ counts = pa.counts;
    const TabG G{pa.G}; const TabLF LF{pa.LF};
    const uint32_t n_shuffles = pa.n_shuffles; const uint64_t seed = pa.seed;
    double* __restrict__ out_p = pa.out_p; uint32_t* __restrict__ out_exceed = pa.out_exceed;
    const uint32_t lane = threadIdx.x & 63u, grp = lane >> 4;
    const uint32_t n_gen = *pa.gen_count;
    if (!pa.gen_count[3]) return;                          // k_perm_fast saw no row that qualifies
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // there is room for the second list when the queue fills at most half of its buffer; else the finished rows are
    // marked in the first list and skipped there.  (Walking a list of mostly finished rows cost k_perm_general one
    // same-address atomic per row: 2.5 ms for the 2e5 rows of the footprint batch, whatever their work.)
    const bool second = 2ull * (unsigned long long)n_gen <= pa.max_rows;
    // (ENUM_ROWS = 16 queued rows per wave and round; with 64 the footprint batch's 2e5 rows are ONE round of the grid:
    //  2.18 ms for this kernel and k_perm_general together against 2.04; 8: 1.95, 32: 2.07)
    for (uint32_t base = blockIdx.x * ENUM_ROWS; base < n_gen; base += gridDim.x * ENUM_ROWS) {
        const uint32_t q_mine = base + lane;
        uint32_t r_mine = 0u, nt_mine = 0u, T_mine[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        bool en_mine = false;
        const bool mine = lane < ENUM_ROWS && q_mine < n_gen;
        if (mine) {
            r_mine = gen_list[q_mine];
#pragma unroll
            for (int k = 0; k < 9; ++k) T_mine[k] = counts[9ull * r_mine + k];
            const uint32_t A0 = T_mine[0] + T_mine[1] + T_mine[2], A1 = T_mine[3] + T_mine[4] + T_mine[5], A2 = T_mine[6] + T_mine[7] + T_mine[8];
            const uint32_t B0 = T_mine[0] + T_mine[3] + T_mine[6], B1 = T_mine[1] + T_mine[4] + T_mine[7], B2 = T_mine[2] + T_mine[5] + T_mine[8];
            // the two smaller row margins and the two smaller column margins (the product of the four ranges does not depend
            // on which of two equal margins is taken as the largest)
            const uint32_t amax = A0 > A1 ? (A0 > A2 ? A0 : A2) : (A1 > A2 ? A1 : A2), bmax = B0 > B1 ? (B0 > B2 ? B0 : B2) : (B1 > B2 ? B1 : B2);
            const uint32_t amin = A0 < A1 ? (A0 < A2 ? A0 : A2) : (A1 < A2 ? A1 : A2), bmin = B0 < B1 ? (B0 < B2 ? B0 : B2) : (B1 < B2 ? B1 : B2);
            const uint32_t amid = A0 + A1 + A2 - amax - amin, bmid = B0 + B1 + B2 - bmax - bmin;
            unsigned long long nt = 1ull;
            const uint32_t ra[2] = {amin, amid}, cb[2] = {bmin, bmid};
            en_mine = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nt *= (unsigned long long)((ra[k >> 1] < cb[k & 1] ? ra[k >> 1] : cb[k & 1]) + 1u);
                en_mine = en_mine && nt <= pa.enum_max;
                if (!en_mine) nt = 1ull;
            }
            en_mine = en_mine && (pa.exact_2x2 || nt <= 4ull * (unsigned long long)n_shuffles);
            nt_mine = (uint32_t)nt;
        }
        if (second) {
            const bool keep = mine && !en_mine;
            const unsigned long long kb = __ballot(keep);
            if (kb) {
                uint32_t at = 0u;
                if (lane == 0) at = atomicAdd(pa.gen_count + 2, (unsigned int)__popcll(kb));
                at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                if (keep) gen_list[n_gen + at + (uint32_t)__popcll(kb & ((1ull << lane) - 1ull))] = r_mine;
            }
        }
        // one row by `width` lanes (16: four rows of the wave at a time, 64: one): L = the lane that read the row
        auto enumerate = [&](int L, bool have, uint32_t width) {
            const uint32_t sub = lane & (width - 1u);
            const int src = have ? L : (int)lane;
            const uint32_t r = (uint32_t)__shfl((int)r_mine, src);
            uint32_t T[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = (uint32_t)__shfl((int)T_mine[k], src);
            unsigned long long mass = 0ull;
            uint32_t ci = 0u, cj = 0u;
            if (have) {
                ci = row_i[r] + pa.site_base; cj = row_j[r] + pa.site_base;
                const uint32_t R0 = T[0] + T[1] + T[2], R1 = T[3] + T[4] + T[5], R2 = T[6] + T[7] + T[8];
                const uint32_t C0 = T[0] + T[3] + T[6], C1 = T[1] + T[4] + T[7], C2 = T[2] + T[5] + T[8];
                const uint32_t N = R0 + R1 + R2;
                uint32_t la = 0u, lb = 0u;                   // the FIRST largest margin
                if (R1 > R0) la = 1u;
                if (R2 > (la ? R1 : R0)) la = 2u;
                if (C1 > C0) lb = 1u;
                if (C2 > (lb ? C1 : C0)) lb = 2u;
                // the table in the layout rows (free, free, la) x columns (free, free, lb)
                const uint32_t pR0 = la == 0u ? R1 : R0, pR1 = la == 2u ? R1 : R2, pR2 = la == 0u ? R0 : la == 1u ? R1 : R2;
                const uint32_t pC0 = lb == 0u ? C1 : C0, pC1 = lb == 2u ? C1 : C2;
                const uint32_t r0 = (pR0 < pC0 ? pR0 : pC0) + 1u, r1 = (pR0 < pC1 ? pR0 : pC1) + 1u;
                const uint32_t r2 = (pR1 < pC0 ? pR1 : pC0) + 1u, r3 = (pR1 < pC1 ? pR1 : pC1) + 1u;
                const uint32_t nt = r0 * r1 * r2 * r3;       // <= enum_max <= 8192 (decided above)
                long long sobs = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) sobs += G[T[k]];
                double c0 = LF[R0];
                c0 += LF[R1]; c0 += LF[R2]; c0 += LF[C0]; c0 += LF[C1]; c0 += LF[C2]; c0 -= LF[N];
                // d / r for d < 2^13, r <= 2^13: the f32 quotient is within one of the integer one
                const float i0 = 1.0f / (float)r0, i1 = 1.0f / (float)r1, i2 = 1.0f / (float)r2;
                auto divmod = [](uint32_t d, uint32_t rr, float inv, uint32_t& rem) {
                    uint32_t q = (uint32_t)((float)d * inv);
                    int m = (int)d - (int)(q * rr);
                    if (m < 0) { q--; m += (int)rr; } else if (m >= (int)rr) { q++; m -= (int)rr; }
                    rem = (uint32_t)m;
                    return q;
                };
                for (uint32_t t = sub; t < nt; t += width) {
                    uint32_t x00, x01, x10, x11;
                    uint32_t d = divmod(t, r0, i0, x00);
                    d = divmod(d, r1, i1, x01);
                    x11 = divmod(d, r2, i2, x10);
                    const int x02 = (int)pR0 - (int)x00 - (int)x01, x12 = (int)pR1 - (int)x10 - (int)x11;
                    const int x20 = (int)pC0 - (int)x00 - (int)x10, x21 = (int)pC1 - (int)x01 - (int)x11;
                    const int x22 = (int)pR2 - x20 - x21;
                    if ((x02 | x12 | x20 | x21 | x22) < 0) continue;
                    const long long ss = G[x00] + G[x01] + G[(uint32_t)x02] + G[x10] + G[x11] + G[(uint32_t)x12] +
                                         G[(uint32_t)x20] + G[(uint32_t)x21] + G[(uint32_t)x22];
                    if (ss < sobs) continue;
                    double lf = LF[x00];
                    lf += LF[x01]; lf += LF[(uint32_t)x02]; lf += LF[x10]; lf += LF[x11]; lf += LF[(uint32_t)x12];
                    lf += LF[(uint32_t)x20]; lf += LF[(uint32_t)x21]; lf += LF[(uint32_t)x22];
                    mass += (unsigned long long)(det_exp(c0 - lf) * 4611686018427387904.0);   // 2^62
                }
            }
            // the sum over the lanes of the group (idle groups shuffle zeros among themselves)
            for (uint32_t o = width >> 1; o > 0u; o >>= 1) mass += __shfl_xor(mass, (int)o);
            if (have) {
                unsigned long long thr = mass >> 30;
                if (thr > 4294967296ull) thr = 4294967296ull;
                if (pa.exact_2x2) {                              // the exact p: the enumerated mass itself
                    if (sub == 0u) {
                        double p = (double)mass * 2.168404344971009e-19;
                        if (p > 1.0) p = 1.0;
                        out_exceed[r] = LGMI_EXCEED_EXACT;
                        out_p[r] = p;
                        if (!second) gen_list[base + (uint32_t)L] = 0xFFFFFFFFu;
                    }
                } else {
                const uint32_t exceed = binom_draw(LF, n_shuffles, thr, ci, cj, k0, k1);   // (every lane of the group the same)
                if (sub == 0u) {
                    out_exceed[r] = exceed;
                    if (out_p) out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
                    if (!second) gen_list[base + (uint32_t)L] = 0xFFFFFFFFu;   // no second list: k_perm_general skips the row
                }
                }
            }
        };
        // rows of up to 256 tables four at a time, sixteen lanes each (a wave per row was bound by the row's chain of
        // dependent look-ups, and the median row — 90 tables — does not fill 64 lanes twice); the larger ones by the whole
        // wave (four of THOSE side by side cost the longest of the four, sixteen lanes wide: measured slower)
        unsigned long long small = __ballot(en_mine && nt_mine <= 256u), big = __ballot(en_mine && nt_mine > 256u);
        while (small) {                                      // (wave-uniform)
            int L = -1;
#pragma unroll
            for (uint32_t g = 0; g < 4u; ++g) {
                if (small) {
                    const int bpos = __ffsll((long long)small) - 1;
                    small &= small - 1ull;
                    if (grp == g) L = bpos;
                }
            }
            enumerate(L, L >= 0, 16u);
        }
        while (big) {
            const int L = __ffsll((long long)big) - 1;
            big &= big - 1ull;
            enumerate(L, true, 64u);
        }
    }
}

// ---------------------------------------------------------------- six-cell tables: exact mass along the perimeter (round 4)
// A 3 x 2 / 2 x 3 table has two degrees of freedom and {S < S_obs} is the lattice inside a convex curve: for a null pair
// at 2e5 reads ~1.4e4 tables in ~100 chords, each chord a 2 x 2 problem.  One more draw changes a hypergeometric CDF by one
// pmf term, so a chord's mass is an affine map of its predecessor's, M_c = rho_c M_p + beta_c, with coefficients made of the
// two chords' bounds, two pmf values and the few values by which the bounds moved: the work is the region's PERIMETER
// (~2e3 lane-steps per row), not its area (1.4e4) and not 1000 rejection-sampled tables (3.1e5 lane-instructions).  The
// exceed count is then one binomial variate, as for 2 x 2 rows.  Specification: six_plan / six_inside_walk in
// oracle/lgmi_perm_oracle.c (canonical form, chord bounds, collapsed table, zero test, gate, the maps and the order they
// are composed in: see the comments there; every name below is theirs).
//   phase S  one lane per queued row: table, canonical form, zero test, chord range from the collapsed table, gate
//   phase C  the wave's qualifying rows one after the other, 64 chords at a time, one lane per chord: two boundary searches
//            from a half-width guess, the map's coefficients, an inclusive scan of the 64 maps, a butterfly sum
//   phase D  one lane per row again: the binomial draws of all the wave's rows side by side
// Rows that stay with the sampling kernel (3 x 3, gate) go to a third list behind the input list, 64 per atomic; when the
// buffer has no room for it (more than a third of all rows queued) the finished rows are marked in the input list instead.
struct SixLists { uint32_t* in; uint32_t n_in; uint32_t* out; bool third; };
__device__ __forceinline__ SixLists six_lists(const PermArgs& pa)
{
    // the rows k_perm_enum left: its second list when it made one, else the whole queue with the finished rows marked
    // (the same test as there); the third list goes behind whichever it is, when there is room
    const uint32_t n_queued = pa.gen_count[0];
    const bool second = pa.enum_max && pa.gen_count[3] && 2ull * (unsigned long long)n_queued <= pa.max_rows;
    SixLists l;
    l.n_in = second ? pa.gen_count[2] : n_queued;
    const uint32_t off = second ? n_queued : 0u;
    l.in = pa.gen_list + off;
    l.third = pa.six_pts && (unsigned long long)off + 2ull * (unsigned long long)l.n_in <= pa.max_rows;
    l.out = pa.gen_list + off + l.n_in;
    return l;
}

__device__ __forceinline__ uint32_t kc22(const HG22& h) {
    // floor(n K / N): the product is below 2^53, so one f64 division lands within one of the quotient (as in k_perm_general)
    const unsigned long long prod = (unsigned long long)h.n * (unsigned long long)h.K;
    uint32_t kc = (uint32_t)((double)prod / (double)h.N);
    const long long rem = (long long)prod - (long long)kc * (long long)h.N;
    if (rem < 0) kc--; else if (rem >= (long long)h.N) kc++;
    if (kc < h.kmin) kc = h.kmin;
    if (kc > h.kmax) kc = h.kmax;
    return kc;
}
// the same with the division replaced by a product with rN ~ 1 / N (any rN within a few ulp: the remainder settles it)
__device__ __forceinline__ uint32_t kc22r(const HG22& h, double rN) {
    const unsigned long long prod = (unsigned long long)h.n * (unsigned long long)h.K;
    uint32_t kc = (uint32_t)((double)prod * rN);
    const long long rem = (long long)prod - (long long)kc * (long long)h.N;
    if (rem < 0) kc--; else if (rem >= (long long)h.N) kc++;
    if (kc < h.kmin) kc = h.kmin;
    if (kc > h.kmax) kc = h.kmax;
    return kc;
}
__device__ __forceinline__ void hg22_set(HG22& h, uint32_t N, uint32_t K, uint32_t n, double c0) {
    h.N = N; h.K = K; h.n = n;
    h.kmin = K + n > N ? K + n - N : 0u;
    h.kmax = K < n ? K : n;
    h.c0 = c0;
}
#ifndef LGMI_SIXABL
#define LGMI_SIXABL 0     // timing-only ablations of k_perm_six (results wrong by construction; tools/abl_six.sh): 1 the maps' sums
                          // and pmf values replaced by constants, 2 the boundary searches replaced by their first guess
#endif
// inside (klo, khi) of {stat22(h, x) < s} around kc: the two monotone boundaries, searched from a half-width guess
__device__ __forceinline__ void inside22(TabG G, const HG22& h, uint32_t kc, long long s, int& klo, int& khi) {
    const long long d0 = s - stat22(G, h, kc);
    int hw = 0;
    if (d0 > 0) {
        // S(x) - S(x*) ~ (x - x*)^2 2^28 / (2 var): a guess only (the search does not depend on it)
        const float Nf = (float)h.N, var = (float)h.n * (float)h.K * ((float)(h.N - h.K) * (float)(h.N - h.n)) / (Nf * Nf * Nf);
        hw = (int)__fsqrt_rn(2.0f * var * (float)d0 * 3.7252903e-09f);
    }
    const int lo = (int)h.kmin - 1, hi = (int)kc, refl = lo + hi;
#if LGMI_SIXABL & 2
    klo = (int)kc - hw < lo ? lo : (int)kc - hw; khi = (int)kc + 1 + hw > (int)h.kmax + 1 ? (int)h.kmax + 1 : (int)kc + 1 + hw; return;
#endif
    // (one Newton step on each guess before the search — two more probes to save the gallop's — was measured slower: 56.6
    //  against 48.3 ms at north-star; the quadratic guess is within a value of the boundary as it is)
    const int gl = (int)kc - hw, gr = (int)kc + 1 + hw;
    klo = refl - first_true32(lo, hi, refl - gl, [&](int j) { return stat22(G, h, (uint32_t)(refl - j)) >= s; });
    khi = first_true32((int)kc + 1, (int)h.kmax + 1, gr, [&](int k) { return stat22(G, h, (uint32_t)k) >= s; });
}

// walk_sum of the oracle: J = pmf22 along lo .. hi (0 outside the support): first = J(lo), last = J(hi), rest = sum over lo + 1 .. hi
__device__ __forceinline__ void walk_sum(TabLF LF, const HG22& h, int lo, int hi, double& first, double& last, double& rest) {
    const int lo2 = lo > (int)h.kmin ? lo : (int)h.kmin, hi2 = hi < (int)h.kmax ? hi : (int)h.kmax;
    first = 0.0; last = 0.0; rest = 0.0;
#if LGMI_SIXABL & 1
    first = 1e-5; last = 1.1e-5; rest = 2e-5 * (double)(hi2 - lo2); return;
#endif
    if (lo2 > hi2) return;
    uint32_t k = (uint32_t)lo2;
    const double term0 = pmf22(LF, h, k);
    double term = term0, sum = 0.0;
    uint32_t rem = (uint32_t)(hi2 - lo2);
#pragma unroll 1
    while (rem > 0u) {
        const uint32_t m = rem < SUB ? rem : SUB;
        double P = 0.0, Nn = 1.0, Q = 1.0;
        const double a = (double)(h.K - k), b = (double)(h.n - k), c = (double)(k + 1u), d = (double)(h.N - h.K - h.n + k + 1u);
        double num = a * b, den = c * d, sn = a + b - 1.0, sd = c + d + 1.0;
#pragma unroll 1
        for (uint32_t j = 0; j < m; ++j) {
            Nn = Nn * num;
            Q = Q * den;
            P = fma(P, den, Nn);
            num -= sn; den += sd; sn -= 2.0; sd += 2.0;
        }
        const double rQ = 1.0 / Q;                       // (one division per sub-block, as unit_sum)
        sum += (term * P) * rQ;
        term = (term * Nn) * rQ;
        k += m;
        rem -= m;
    }
    if (lo >= (int)h.kmin) { first = term0; rest = sum; } else { rest = term0 + sum; }
    if (hi <= (int)h.kmax) last = term;
}

#ifndef LGMI_SIX_WPS
#define LGMI_SIX_WPS 7     // waves per SIMD the register budget is set for.  With the row figures and the scans still going through
                           // ds_bpermute: 4 (108 VGPRs) 46.1 ms, 5 (96 + 10 spilled) 41.7, 6 (80 + 38) 44.2.  With DPP scans and the row
                           // figures read from LDS (5: 96, none spilled, 41.2 ms): 6 (80 + 10) 39.0, 7 (72 + 18) 37.8, 8 (64 + 31) 44.2; with the row's own
                           // state parked in LDS across the trips: 7 (72 + 10) 37.7, 8 (64 + 20) 38.1
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LGMI_SIX_WPS, LGMI_SIX_WPS))) void k_perm_six(PermArgs pa)
{
    const uint32_t* __restrict__ row_i = pa.row_i; const uint32_t* __restrict__ row_j = pa.row_j;
    const uint32_t* __restrict__ counts = pa.counts;
    const TabG G{pa.G}; const TabLF LF{pa.LF};
    const uint32_t n_shuffles = pa.n_shuffles; const uint64_t seed = pa.seed;
    double* __restrict__ out_p = pa.out_p; uint32_t* __restrict__ out_exceed = pa.out_exceed;
    const uint32_t lane = threadIdx.x & 63u;
    const SixLists ls = six_lists(pa);
    unsigned long long box_max = ((unsigned long long)pa.six_pts * (unsigned long long)n_shuffles) >> 4;
    if (box_max > 4194304ull) box_max = 4194304ull;
    if (pa.exact_2x2) box_max = 1048576ull;                 // the exact-p mode: whatever the walk reaches, with or without shuffles
    __shared__ unsigned long long s_acc[64];                // the rows' inside masses (2^-62, integer sums)
    __shared__ __attribute__((aligned(16))) uint32_t s_row[64 * 16];   // the round's rows: what a chord needs of its row
    unsigned int* const next_row = pa.gen_count + 5;
    uint32_t q_next = 0u;
    if (lane == 0) q_next = atomicAdd(next_row, 64u);
    for (;;) {                                               // every wave reaches q0 >= n_in: the grid drains
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_next);
        if (q0 >= ls.n_in) break;
        if (lane == 0) q_next = atomicAdd(next_row, 64u);
        // ---- phase S
        const uint32_t q = q0 + lane;
        uint32_t r = 0xFFFFFFFFu;
        if (q < ls.n_in) r = ls.in[q];
        int st = 0;                                          // 0 nothing, 1 stays with k_perm_general, 2 six-cell with chords, 3 six-cell, thr known
        uint32_t Ao = 0u, Ap = 1u, Aq = 1u, B0 = 0u, nz = 0u;
        int zlo = 0;
        long long sobs = 0;
        double cJ = 0.0, rN = 0.0;
        unsigned long long ins_x = 0ull;                     // rows decided here: their inside mass (2^-62)
        if (r != 0xFFFFFFFFu) {
            uint32_t T[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) T[k] = counts[9ull * r + k];
            const uint32_t R0 = T[0] + T[1] + T[2], R1 = T[3] + T[4] + T[5], R2 = T[6] + T[7] + T[8];
            const uint32_t C0 = T[0] + T[3] + T[6], C1 = T[1] + T[4] + T[7], C2 = T[2] + T[5] + T[8];
            const uint32_t N = R0 + R1 + R2;
            const int nr = (R0 != 0u) + (R1 != 0u) + (R2 != 0u), nc = (C0 != 0u) + (C1 != 0u) + (C2 != 0u);
            st = 1;
            if (nr * nc == 6) {
                uint32_t A0, A1, A2, B1;
                if (nr == 3) { A0 = R0; A1 = R1; A2 = R2; B0 = C0 ? C0 : C1; B1 = C0 ? (C1 ? C1 : C2) : C2; }
                else { A0 = C0; A1 = C1; A2 = C2; B0 = R0 ? R0 : R1; B1 = R0 ? (R1 ? R1 : R2) : R2; }
                int o = 0;
                uint32_t am = A0;
                if (A1 < am) { o = 1; am = A1; }
                if (A2 < am) { o = 2; am = A2; }
                Ao = am;
                Ap = o == 0 ? A1 : A0;
                Aq = o == 2 ? A1 : A2;
                rN = 1.0 / (double)(Ap + Aq);
#pragma unroll
                for (int k = 0; k < 9; ++k) sobs += G[T[k]];
                cJ = LF[A0];
                cJ += LF[A1]; cJ += LF[A2]; cJ += LF[B0]; cJ += LF[B1]; cJ -= LF[N];
                double b = cJ;
                b += (double)N;
                b += det_log(((double)Ao + 1.0) * ((double)Ap + 1.0));
                b -= (double)sobs * 3.725290298461914e-09;
                if (b < -23.1) {                                                     // (thr = 0)
                    st = 3; ins_x = 4611686018427387904ull;
                    if (pa.exact_2x2) {
                        // the exact p of such a row is the bound itself (an upper bound of the tail mass, never 0.0; six_thr in
                        // the specification): carried to phase D as the double's bits under bit 63, which no mass ever sets
                        const double pb = det_exp(b);
                        ins_x = (unsigned long long)__double_as_longlong(pb > 2.2250738585072014e-308 ? pb : 2.2250738585072014e-308) | (1ull << 63);
                    }
                }
                else {
                    HG22 hc;
                    hg22_set(hc, N, Ao, B0, 0.0);
                    const long long sc = sobs + 8 - (G[Ap] + G[Aq] - G[Ap + Aq]);
                    const uint32_t kcc = kc22(hc);
                    int zhi;
                    inside22(G, hc, kcc, sc, zlo, zhi);
                    if (zhi - zlo - 1 <= 0) { st = 3; ins_x = 0ull; }               // (thr = 2^32)
                    else {
                        int zc = (int)kcc;
                        if (zc <= zlo) zc = zlo + 1;
                        if (zc >= zhi) zc = zhi - 1;
                        HG22 h;
                        hg22_set(h, Ap + Aq, Ap, B0 - (uint32_t)zc, 0.0);
                        int klo, khi;
                        inside22(G, h, kc22r(h, rN), sobs - G[(uint32_t)zc] - G[Ao - (uint32_t)zc], klo, khi);
                        const int len = khi - klo - 1;
                        nz = (uint32_t)(zhi - zlo - 1);
                        const unsigned long long box = (unsigned long long)nz + (unsigned long long)(len > 1 ? len : 1);
                        if (box <= box_max) st = 2;
                    }
                }
            }
        }
        // rows that stay with k_perm_general: the third list, one atomic per wave and round
        {
            const unsigned long long kb = __ballot(st == 1);
            if (kb && ls.third) {
                uint32_t at = 0u;
                if (lane == 0) at = atomicAdd(pa.gen_count + 4, (unsigned int)__popcll(kb));
                at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                if (st == 1) ls.out[at + (uint32_t)__popcll(kb & ((1ull << lane) - 1ull))] = r;
            }
        }
        // ---- phase C: the perimeter walk (six_inside_walk in the oracle: every chord an affine map of its predecessor's
        //      mass, composed 16 chords at a time).  The sub-chunks of the wave's rows are laid side by side, four per trip:
        //      a chord's composite depends on its distance from its sub-chunk's start only, the first chord of a row has
        //      rho = 0 (whatever sits in the lanes before it drops out exactly), and the masses are added as integers.
        {
            const uint32_t my_sc = st == 2 ? (nz + 15u) >> 4 : 0u;
            uint32_t incl = my_sc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_up(incl, o);
                if (lane >= (uint32_t)o) incl += v;
            }
            const uint32_t my_pre = incl - my_sc;            // sub-chunks before this lane's row
            __syncthreads();                                 // (one wave per workgroup) the previous round's reads are over
            s_acc[lane] = st == 3 ? ins_x : 0ull;            // (rows decided in phase S: their inside mass as it is; nothing is added)
            {   // the row's figures where every chord's lane can read them (four wide LDS reads a trip; fifteen ds_bpermute before)
                uint4* const q = reinterpret_cast<uint4*>(s_row + lane * 16u);
                const unsigned long long sb = (unsigned long long)sobs, cb = (unsigned long long)__double_as_longlong(cJ), nb = (unsigned long long)__double_as_longlong(rN);
                q[0] = make_uint4(Ao, Ap, Aq, B0);
                q[1] = make_uint4(nz, (uint32_t)zlo, my_pre, r);             // (.w and q[3].z: this lane's own row and state, read back

                q[2] = make_uint4((uint32_t)sb, (uint32_t)(sb >> 32), (uint32_t)cb, (uint32_t)(cb >> 32));
                q[3] = make_uint4((uint32_t)nb, (uint32_t)(nb >> 32), (uint32_t)st, 0u);   //  in phase D — six registers fewer across the trips)
            }
            __syncthreads();
            const uint32_t TS = bcast32(incl, 63);           // wave-uniform
            const uint32_t grp = lane >> 4, sub = lane & 15u;
            double M_it = 0.0;                               // carried from lane 63 of the previous trip
            int a_it = 0, b_it = 0;
            for (uint32_t sid0 = 0u; sid0 < TS; sid0 += 4u) {
                const uint32_t sid = sid0 + grp;
                const bool valid = sid < TS;
                // the row of sub-chunk sid: the largest slot with pre[slot] <= sid — pre[] does not decrease along the lanes, so
                // it is the number of lanes with pre <= sid, minus one: a ballot per group instead of a search in LDS
                uint32_t rs_u = 0u, pre_rs = 0u;
#pragma unroll
                for (uint32_t g = 0; g < 4u; ++g) {
                    const unsigned long long le = __ballot(my_pre <= sid0 + g);      // (lane 0 holds pre = 0: never empty)
                    const uint32_t slot = (uint32_t)__popcll(le) - 1u;
                    if (grp == g) rs_u = slot;
                }
                const int rs = (int)rs_u;
                const uint4* const q = reinterpret_cast<const uint4*>(s_row + rs_u * 16u);
                const uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                pre_rs = q1.z;
                const uint32_t rAo = q0.x, rAp = q0.y, rAq = q0.z, rB0 = q0.w, rnz = q1.x;
                const int rzlo = (int)q1.y;
                const long long rsobs = (long long)(((unsigned long long)q2.y << 32) | q2.x);
                const double rcJ = __longlong_as_double((long long)(((unsigned long long)q2.w << 32) | q2.z));
                const double rrN = __longlong_as_double((long long)(((unsigned long long)q3.y << 32) | q3.x));
                const uint32_t j = valid ? 16u * (sid - pre_rs) + sub : 0u;
                const bool active = valid && j < rnz;
                const uint32_t Np = rAp + rAq, K = rAp;
                HG22 h = {Np, K, 0u, 0u, 0u, 0.0};
                int a = 0, b = 0, kc = 0;
                uint32_t z = 0u;
                if (active) {
                    z = (uint32_t)(rzlo + 1) + j;
                    double c0 = rcJ;
                    c0 -= LF[z];
                    c0 -= LF[rAo - z];
                    hg22_set(h, Np, K, rB0 - z, c0);
                    kc = (int)kc22r(h, rrN);
                    int klo, khi;
                    inside22(G, h, (uint32_t)kc, rsobs - G[z] - G[rAo - z], klo, khi);
                    a = klo; b = khi - 1;
                }
                int a_p = __builtin_amdgcn_update_dpp(a, a, 0x138, 0xF, 0xF, false),     // wave_shr:1 (lane 0: see below)
                    b_p = __builtin_amdgcn_update_dpp(b, b, 0x138, 0xF, 0xF, false);
                if (lane == 0u) { a_p = a_it; b_p = b_it; }
                const bool first = j == 0u;
                if (first) { a_p = kc; b_p = kc; }
                double rho = 1.0, beta = 0.0;                // idle lanes: the identity map
                if (active) {
                    double fT, lT, rT, fB, lB, rB;
                    walk_sum(LF, h, b < b_p ? b : b_p, b < b_p ? b_p : b, fT, lT, rT);
                    walk_sum(LF, h, a < a_p ? a : a_p, a < a_p ? a_p : a, fB, lB, rB);
                    const double adjT = b >= b_p ? rT : -rT, adjB = a < a_p ? rB : -rB;
                    double t2 = 0.0;
                    rho = 0.0;
                    if (!first) {
                        const double Jb = b >= b_p ? fT : lT, Ja = a < a_p ? lB : fB;
                        const double zd = (double)z;
                        const double rzd = 1.0 / (zd * (double)(Np - h.n));
                        const double f = (double)(rAo - z + 1u) * (double)(h.n + 1u);
                        rho = f * rzd;
                        const double tb = Jb * (double)((int)K - b_p), ta = Ja * (double)((int)K - a_p);
                        t2 = tb - ta;
                        t2 = t2 * (zd * rzd);
                    }
                    beta = t2 + adjT;
                    beta = beta + adjB;
                    if (b - a <= 0) { rho = 0.0; beta = 0.0; }
                }
                // inclusive scan of the maps inside each sub-chunk (Hillis-Steele).  A sub-chunk is one 16-lane DPP row: the
                // neighbour's value comes by row_shr (a VALU move) instead of ds_bpermute (an LDS round trip per step)
#define LGMI_SCAN_STEP(O)                                                                                   \
                {                                                                                             \
                    const double pr = dpp_row_shr_f64<O>(rho), pb = dpp_row_shr_f64<O>(beta);                   \
                    if (sub >= (uint32_t)(O)) {                                                               \
                        const double x = rho * pb;                                                            \
                        beta = x + beta;                                                                      \
                        rho = rho * pr;                                                                       \
                    }                                                                                         \
                }
                LGMI_SCAN_STEP(1) LGMI_SCAN_STEP(2) LGMI_SCAN_STEP(4) LGMI_SCAN_STEP(8)
#undef LGMI_SCAN_STEP
                // the mass carried into each sub-chunk: the last chord of the sub-chunk before it (the same row's, or dropped by
                // rho = 0 when a row starts here)
                double carry = M_it, M = 0.0;
#pragma unroll
                for (uint32_t g = 0; g < 4u; ++g) {
                    double x = rho * carry;
                    x = x + beta;
                    if (grp == g) M = x;
                    carry = __longlong_as_double((long long)bcast64((unsigned long long)__double_as_longlong(x), (int)(16u * g + 15u)));
                }
                M_it = carry;
                a_it = (int)bcast32((uint32_t)a, 63); b_it = (int)bcast32((uint32_t)b, 63);
                // (adding the sub-chunk's masses inside its 16 lanes first, one atomic per sub-chunk, was no faster: 48.8 against 48.3 ms)
                if (active && M > 0.0)
                    atomicAdd(&s_acc[rs], M >= 1.0 ? 4611686018427387904ull : (unsigned long long)(M * 4611686018427387904.0));
            }
            __syncthreads();
        }
        // ---- phase D: thr = (2^62 - inside) >> 30 for every six-cell row (inside = 2^62 after the zero test: 0; inside = 0 for
        //      an empty chord range: 2^32)
        const uint32_t r_d = s_row[lane * 16u + 7u];
        const bool fin = s_row[lane * 16u + 14u] >= 2u;          // state 2 (walked) or 3 (decided in phase S)
        if (fin) {
            const unsigned long long ins = s_acc[lane];
            if (pa.exact_2x2) {                              // the exact p: one minus the inside mass (six_thr's p_out in the oracle)
                double p = 1.0 - (double)ins * 2.168404344971009e-19;
                if (p < 0.0) p = 0.0;
                if (ins >> 63) p = __longlong_as_double((long long)(ins & ~(1ull << 63)));   // the zero test's bound (phase S)
                out_exceed[r_d] = LGMI_EXCEED_EXACT;
                out_p[r_d] = p;
            } else {
                const unsigned long long thr = ins <= 4611686018427387904ull ? (4611686018427387904ull - ins) >> 30 : 0ull;
                const uint32_t exceed = binom_draw(LF, n_shuffles, thr, row_i[r_d] + pa.site_base, row_j[r_d] + pa.site_base, (uint32_t)seed, (uint32_t)(seed >> 32));
                out_exceed[r_d] = exceed;
                if (out_p) out_p[r_d] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
            }
            if (!ls.third) ls.in[q0 + lane] = 0xFFFFFFFFu;  // no third list: k_perm_general skips the row
        }
        const uint32_t done = (uint32_t)__popcll(__ballot(fin));           // (statistics)
        if (lane == 0 && done) atomicAdd(pa.gen_count + 6, done);
    }
}

// ---------------------------------------------------------------- general tables
//
// One wave per queued row; lane l runs shuffles l, l+64, ...  A shuffle is a chain of up to
// four conditional hypergeometric draws, each a rejection loop of random length, so a
// lock-step "draw by draw" loop would make every draw cost the slowest lane's retries.
// Instead every lane runs its own state machine and one trip of the wave loop is ONE
// candidate (HRUA) or ONE urn step for whatever draw the lane is in; a lane whose draw
// finishes resolves the following trivially-determined draws, starts its next shuffle if
// needed and sets up the next real draw inside the same trip.  The stream of uniforms of a
// shuffle is consumed strictly in order, so the result is that of the sequential
// specification (oracle/lgmi_perm_oracle.c: perm_one) whatever the interleaving.
#define HRUA_D1 1.7155277699214135
#define HRUA_D2 0.8989161620588988

// the three quotients of a (pop, good), as products with the reciprocals of pop, pop - 1, pop + 2: most draws of
// a row share the population size, so the divisions happen once per row, not once per draw
struct HrRecip { uint32_t pop; double rp, rp1, rp2; };
struct HrBase { uint32_t pop, good; double d4, cvar, c9; };

__device__ __forceinline__ void hr_recip(uint32_t pop, HrRecip& r) {
    r.pop = pop;
    r.rp = 1.0 / (double)pop;
    r.rp1 = 1.0 / (double)(pop - 1u);
    r.rp2 = 1.0 / ((double)pop + 2.0);
}
__device__ __forceinline__ void hr_base(const HrRecip& r, uint32_t good, HrBase& b) {      // r holds the reciprocals of b's pop
    const uint32_t bad = r.pop - good, mn = good < bad ? good : bad;
    b.pop = r.pop; b.good = good;
    b.d4 = (double)mn * r.rp;
    b.cvar = b.d4 * (1.0 - b.d4) * r.rp1;
    b.c9 = (double)(mn + 1u) * r.rp2;
}

struct GState {
    // Philox call index inside the current shuffle
    uint32_t call;
    // table being drawn
    uint32_t s, rr0, rr1, pop_all, cc, pop, xa;
    uint32_t sp1, sp2, sp3;   // words 1 - 3 of the last Philox call: two pairs of 24-bit uniforms (next_pair in the oracle)
    int spare;                // how many of the two pairs are still unused
    int aft;                  // the last draw was the threshold-table draw (the row's hat width applies to the next one)
    int d;                 // draw index: column d>>1, row d&1
    long long ss;
    // draw in flight
    int phase;             // 1 HRUA candidate loop, 2 urn loop, 3 lane finished
    uint32_t good, sample, m, mn, mx;
    double d6, d8, d10, d11;
    uint32_t rem_total, rem_good, left;
};

// Per-row state in LDS (one wave per workgroup, so __syncthreads() is a wave barrier):
//   tab_thr[e]   inverse-CDF thresholds of the first real draw of a shuffle.  Its parameters are the same in every
//                shuffle of the row, so it is drawn with one 32-bit word and a binary search instead of a
//                rejection loop (window of at most FIRST_MAX values around the mode, else the HRUA path stays)
//   tab_guide[b] where the search starts for a word whose top GUIDE_BITS bits are b (256 buckets: 1 - 3 probes instead of
//                11; round 3 measured 1024 and 2048 buckets — 0 - 1 probes — within 1 % of 256: tools/exp_guide.sh)
//   next_s       the next shuffle index nobody has taken: a lane that finishes a shuffle takes the next one, so
//                the wave drains together whatever the lanes' rejection counts were (the exceed count is a sum
//                over shuffles and every shuffle has its own Philox stream: who runs which one does not matter)
#ifndef LGMI_FIRST_MAX
#define LGMI_FIRST_MAX 2032
#endif
static const uint32_t FIRST_MAX = LGMI_FIRST_MAX;
#ifndef LGMI_GUIDE_BITS
#define LGMI_GUIDE_BITS 8
#endif
static const uint32_t GUIDE_SH = 32u - LGMI_GUIDE_BITS, GUIDE_N = 1u << LGMI_GUIDE_BITS;
static const uint32_t XRING = 512;
// Lock-step rows: what the set-up of a second draw looks up depends on the first result x0 alone — the log-weight at the
// mode (four LF) and the G of the two cells x0 fixes — and the first results of a row pile up around their mode.  The
// window entries that expect to be drawn at least once in n_shuffles draws (mode +- sd sqrt(2 ln(n_shuffles / 2.5 sd)))
// get the two sums once per row, 16 bytes each, in the part of the threshold table's LDS the row's window leaves free
// (a 5 % third allele at 2e5 reads: ~600 of the 2032 words are thresholds, room for 350 entries); a shuffle whose x0 falls
// there reads LDS instead of making six scattered look-ups.  Same values, same order of the additions: nothing changes
// in the specification, and which entries are cached changes no result (so the range may come from approximate math).
// Why: the kernel is bound by the L1's rate of scattered look-ups — 15 per table draw before this
// (profiles/r03_pmc_perm_general.json); a fixed 256-entry cache of its own cost occupancy (11 workgroups per CU) and
// still took north-star from 280 to 243 ms, cfg5 from 1786 to 1448 ms (profiles/r03_perm_general_cache.txt).
#ifndef LGMI_LS_CACHE
#define LGMI_LS_CACHE 1            // 0: no cache (timing comparisons)
#endif      // entries of the lock-step rows' ring of first draws (a power of two)

// rint(p * 2^52) for 0 <= p < 1 without a 64-bit conversion: adding 2^52 leaves the rounded value in the mantissa
__device__ __forceinline__ unsigned long long to_fixed52(double p) {
    const double t = p * 4503599627370496.0 + 4503599627370496.0;
    return (unsigned long long)__double_as_longlong(t) & 0x000FFFFFFFFFFFFFull;
}

#ifndef LGMI_PERM_WPS
#define LGMI_PERM_WPS 4            // waves per SIMD the register budget of k_perm_general is set for
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LGMI_PERM_WPS, LGMI_PERM_WPS))) void k_perm_general(PermArgs pa)
{
    const unsigned int* __restrict__ gen_count = pa.gen_count;
    const uint32_t* __restrict__ row_i = pa.row_i; const uint32_t* __restrict__ row_j = pa.row_j;
    const uint32_t* __restrict__ counts = pa.counts;
    const TabG G{pa.G}; const TabLF LF{pa.LF};
    const uint32_t n_shuffles = pa.n_shuffles; const uint64_t seed = pa.seed;
    double* __restrict__ out_p = pa.out_p; uint32_t* __restrict__ out_exceed = pa.out_exceed;
    __shared__ __attribute__((aligned(16))) uint32_t tab_thr[FIRST_MAX];   // thresholds [0, tab_n), then the lock-step rows' cache
    __shared__ uint16_t tab_guide[GUIDE_N + 1u];   // tab_guide[b] = the draw for u = b << GUIDE_SH: where the search for u >> GUIDE_SH == b starts
    __shared__ uint32_t next_s;
    __shared__ __attribute__((aligned(8))) uint16_t x_ring[XRING];   // lock-step rows: first draws waiting for a lane
    const uint32_t lane = threadIdx.x & 63u;
    // the rows k_perm_six left in its third list when it made one; else the rows k_perm_enum left: its second list when it
    // made one, else the whole queue with the finished rows marked (six_lists: the same tests as in those kernels)
    const SixLists sl = six_lists(pa);
    const uint32_t n_gen = sl.third ? gen_count[4] : sl.n_in;
    const uint32_t* __restrict__ gen_list = sl.third ? sl.out : sl.in;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // rows are taken from a shared counter (gen_count[1], zero at launch), the next one asked for while the current
    // one runs: rows differ in cost (table size, streamlined or general loop) and a fixed stride left the average
    // wave idle for the last ~15 % of the kernel (SQ_WAVE_CYCLES against SQ_BUSY_CYCLES)
    unsigned int* const next_row = pa.gen_count + 1;
    // QBATCH rows per atomic (same-address atomics are served at ~60 M/s chip-wide); one at a time when there are so
    // few rows that the batch itself would unbalance the waves
    const uint32_t batch = n_gen >= 64u * gridDim.x ? QBATCH : 1u;
    uint32_t q_next = 0u;
    if (lane == 0) q_next = atomicAdd(next_row, batch);
    for (;;) {                                           // every wave reaches q0 >= n_gen: the grid drains
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q_next);
        if (q0 >= n_gen) break;
        if (lane == 0) q_next = atomicAdd(next_row, batch);
    for (uint32_t qk = 0; qk < batch && qk < n_gen - q0; ++qk) {
        const uint32_t q = q0 + qk;
        const uint32_t r = gen_list[q];
        if (r == 0xFFFFFFFFu) continue;                  // done by k_perm_enum (wave-uniform)
        const uint32_t ci = row_i[r] + pa.site_base, cj = row_j[r] + pa.site_base;
        uint32_t T[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) T[k] = counts[9ull * r + k];
        long long sobs = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) sobs += G[T[k]];
        // compact margins: non-empty rows / columns first, order kept.  Empty rows and columns and
        // the last non-empty row / column are fully determined (their draws consume no random
        // numbers in the sequential specification), so only (nr-1) x (nc-1) draws are real.
        uint32_t Rr[3] = {T[0] + T[1] + T[2], T[3] + T[4] + T[5], T[6] + T[7] + T[8]};
        uint32_t Cq[3] = {T[0] + T[3] + T[6], T[1] + T[4] + T[7], T[2] + T[5] + T[8]};
        const uint32_t N = Rr[0] + Rr[1] + Rr[2];
        uint32_t R0 = 0, R1 = 0, R2 = 0, C0 = 0, C1 = 0;
        int nr = 0, nc = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (Rr[a]) { if (nr == 0) R0 = Rr[a]; else if (nr == 1) R1 = Rr[a]; else R2 = Rr[a]; nr++; }
            if (Cq[a]) { if (nc == 0) C0 = Cq[a]; else if (nc == 1) C1 = Cq[a]; nc++; }
        }
        HrRecip rcp;           // reciprocals of the population size last seen
        rcp.pop = 0u; rcp.rp = 0.0; rcp.rp1 = 0.0; rcp.rp2 = 0.0;

        // ---- per-row threshold table of the first draw (wave-uniform control flow; integer prefix sums, so the
        //      result does not depend on the order of the additions — see the CPU specification)
        bool tab_ok = false;
        uint32_t tab_klo = 0, tab_n = 0, tab_mode = 0;
        float tab_sd = 1.0f;
        {
            const uint32_t pop = N, good = R0, sample = C0;
            const uint32_t m = sample < pop - sample ? sample : pop - sample;
            if (m >= 10u) {
                const uint32_t kmin = sample + good > pop ? sample + good - pop : 0u, kmax = good < sample ? good : sample;
                const double pg = (double)good / (double)pop;
                const double var = (double)sample * pg * (1.0 - pg) * (double)(pop - sample) / (double)(pop - 1u);
                const double sd = det_sqrt(var + 1.0);
                const uint32_t w = (uint32_t)floor(6.5 * sd) + 4u;
                // floor((sample+1)(good+1) / (pop+2)): the product is below 2^53, so one f64 division lands within one
                // of the integer quotient and an integer remainder settles it (a 64-bit integer division costs 90 instructions)
                const unsigned long long prod = (unsigned long long)(sample + 1u) * (unsigned long long)(good + 1u);
                uint32_t mode = (uint32_t)((double)prod / ((double)pop + 2.0));
                {
                    const long long rem = (long long)prod - (long long)mode * (long long)(pop + 2u);
                    if (rem < 0) mode--; else if (rem >= (long long)(pop + 2u)) mode++;
                }
                if (mode < kmin) mode = kmin;
                if (mode > kmax) mode = kmax;
                tab_klo = mode - kmin > w ? mode - w : kmin;
                tab_mode = mode;
                tab_sd = (float)sd;
                const uint32_t khi = kmax - mode > w ? mode + w : kmax;
                tab_n = khi - tab_klo + 1u;
                tab_ok = tab_n <= FIRST_MAX;
            }
            __syncthreads();                           // the previous row's table reads are over
            if (lane == 0) next_s = 64u;
            if (tab_ok) {
                double c0 = LF[good];
                c0 += LF[pop - good];
                c0 += LF[sample];
                c0 += LF[pop - sample];
                c0 -= LF[pop];
                const uint32_t seg = (tab_n + 63u) / 64u;
                const uint32_t e0 = lane * seg < tab_n ? lane * seg : tab_n;
                const uint32_t e1 = e0 + seg < tab_n ? e0 + seg : tab_n;
                unsigned long long loc = 0ull;         // running sum of pmf * 2^52 inside the lane's segment
                double pm = 0.0;
                if (e0 < e1) {                         // first entry of the segment: from the log-factorials
                    const uint32_t k = tab_klo + e0;
                    double x = c0;
                    x -= LF[k];
                    x -= LF[good - k];
                    x -= LF[sample - k];
                    x -= LF[pop - good - sample + k];
                    pm = det_exp(x);
                    loc = to_fixed52(pm);
                    tab_thr[e0] = (uint32_t)(loc >> 20);
                }
                if (e0 + 1u < e1) {                     // the others: by the hypergeometric ratio num / den, with
                    // num = (good-k+1)(sample-k+1) and den = k (pop-good-sample+k) stepped by their second differences:
                    // integers below 2^53, so the same doubles as the products (as in unit_mass)
                    const uint32_t k = tab_klo + e0 + 1u;
                    const double a = (double)(good - k + 1u), b = (double)(sample - k + 1u), c = (double)k, d = (double)(pop - good - sample + k);
                    double num = a * b, den = c * d, sn = a + b - 1.0, sd = c + d + 1.0;
                    for (uint32_t e = e0 + 1u; e < e1; ++e) {
                        pm = pm * num / den;
                        loc += to_fixed52(pm);
                        tab_thr[e] = (uint32_t)(loc >> 20);
                        num -= sn; den += sd; sn -= 2.0; sd += 2.0;
                    }
                }
                unsigned long long incl = loc;         // inclusive scan of the segment totals over the lanes
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned long long v = __shfl_up(incl, o);
                    if (lane >= (uint32_t)o) incl += v;
                }
                const unsigned long long before = (incl - loc) >> 20;
                for (uint32_t b = lane; b < GUIDE_N + 1u; b += 64u) tab_guide[b] = (uint16_t)(tab_n - 1u);
                // the final threshold just before this lane's segment (the last one of the lane below; segments are
                // dealt in lane order, so a lane with entries has a full lane below it)
                const unsigned long long my_last = (loc >> 20) + before;
                const uint32_t last_final = my_last >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)my_last;
                const uint32_t below = __shfl_up(last_final, 1);
                __syncthreads();                       // guide defaults written before any bucket is
                // one pass: add what the lanes below sum to, clamp, and let entry e — the answer for every u in
                // [thr[e-1], thr[e]) — start the buckets whose first value b << GUIDE_SH falls in that range (thresholds do
                // not decrease, so each bucket is written once)
                uint32_t prev = lane > 0u && e0 < tab_n ? below : 0u;
                for (uint32_t e = e0; e < e1; ++e) {
                    const unsigned long long t = (unsigned long long)tab_thr[e] + before;
                    const uint32_t cur = t >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
                    tab_thr[e] = cur;
                    if (cur > prev) {
                        const uint32_t b_hi = (cur - 1u) >> GUIDE_SH;
                        for (uint32_t b = (uint32_t)(((unsigned long long)prev + ((1ull << GUIDE_SH) - 1ull)) >> GUIDE_SH); b <= b_hi; ++b)
                            tab_guide[b] = (uint16_t)e;
                    }
                    prev = cur;
                }
            }
            __syncthreads();
        }

        // ---- 3 x 2 / 2 x 3 rows: one HRUA hat width for the draw that follows the table draw, the largest over the
        //      first draw's window (any width >= Stadlober's keeps HRUA exact; CPU specification: hrua_width_bound)
        double d7max = 0.0;
        uint32_t key_pop2 = 0u, key2 = 0u;
        if (tab_ok && nr * nc == 6 && N - R0 > 1u && N - C0 > 1u) {
            const uint32_t xlo = tab_klo, xhi = tab_klo + tab_n - 1u;
            double varmax;
            if (nr == 3) {
                const uint32_t pop2 = N - R0, good = R1, bad = pop2 - good, mn = good < bad ? good : bad, half = pop2 / 2u;
                const uint32_t s_lo = C0 - xhi, s_hi = C0 - xlo;
                const uint32_t m_lo = s_lo < pop2 - s_lo ? s_lo : pop2 - s_lo, m_hi = s_hi < pop2 - s_hi ? s_hi : pop2 - s_hi;
                const uint32_t m_max = (s_lo <= half && half <= s_hi) ? half : (m_lo > m_hi ? m_lo : m_hi);
                const double rp = 1.0 / (double)pop2, rp1 = 1.0 / (double)(pop2 - 1u);
                const double d4 = (double)mn * rp, cvar = d4 * (1.0 - d4) * rp1;
                varmax = (double)(pop2 - m_max) * (double)m_max * cvar;
                key_pop2 = pop2; key2 = good;
            } else {
                const uint32_t pop2 = N - C0, m = C1 < pop2 - C1 ? C1 : pop2 - C1, half = pop2 / 2u;
                const uint32_t g_lo = R0 - xhi, g_hi = R0 - xlo;
                const uint32_t n_lo = g_lo < pop2 - g_lo ? g_lo : pop2 - g_lo, n_hi = g_hi < pop2 - g_hi ? g_hi : pop2 - g_hi;
                const uint32_t mn_max = (g_lo <= half && half <= g_hi) ? half : (n_lo > n_hi ? n_lo : n_hi);
                const double rp = 1.0 / (double)pop2, rp1 = 1.0 / (double)(pop2 - 1u);
                const double d4 = (double)mn_max * rp, cvar = d4 * (1.0 - d4) * rp1;
                varmax = (double)(pop2 - m) * (double)m * cvar;
                key_pop2 = pop2; key2 = C1;
            }
            d7max = det_sqrt(varmax + 0.5) * (1.0 + 9.094947017729282e-13);   // 1 + 2^-40 against rounding
        }

        // ---- 3 x 2 and 2 x 3 tables whose two real draws are "threshold table, then HRUA" for every possible
        //      first result (the usual case: a tri-allelic site against a bi-allelic one at hundreds of reads or
        //      more): the 64 lanes are the 64 LOCK-STEP CANDIDATE STREAMS of the specification (perm_lockstep in
        //      oracle/lgmi_perm_oracle.c).  The first draws X[k] are made in bulk through the row's table, four per
        //      Philox call with every lane busy, into a ring in LDS; a trip is one HRUA candidate per lane — a Philox
        //      call per lane serves two trips, no lane ever needs a first draw inside the loop — and the lanes that
        //      accept score their table (the statistic in closed form), take the next first draws in lane order
        //      (a ballot, no atomics) and set up their next second draw.  The lanes stop together after the trip in
        //      which the n_shuffles-th table is scored: no drain tail.  Anything else (urn-sized draws, draws that
        //      can be trivially determined, 3 x 3) takes the general path below.
        bool simple = false;
        if (tab_ok && nr * nc == 6 && d7max > 0.0) {
            const uint32_t xlo = tab_klo, xhi = tab_klo + tab_n - 1u;
            if (nr == 3) {
                const uint32_t pop2 = N - R0;
                simple = (C0 - xhi >= 10u) && (pop2 - (C0 - xlo) >= 10u) && (C0 - xlo < pop2);
            } else {
                const uint32_t pop2 = N - C0, m2 = C1 < pop2 - C1 ? C1 : pop2 - C1;
                simple = (m2 >= 10u) && (R0 - xhi >= 1u) && (R0 - xlo < pop2);
            }
        }
        if (simple) {
            const uint32_t pop2 = nr == 3 ? N - R0 : N - C0;
            HrRecip rc2;                                 // the second draw's population is the row's; nr == 3: its good too
            hr_recip(pop2, rc2);
            HrBase hb;
            hb.pop = pop2; hb.good = 0u; hb.d4 = 0.0; hb.cvar = 0.0; hb.c9 = 0.0;
            if (nr == 3) hr_base(rc2, R1, hb);
            const double d8 = HRUA_D1 * d7max + HRUA_D2, lim16 = 16.0 * d7max;   // the row's hat (no square root per shuffle)
            uint32_t filled = 0u, next = 0u, total = 0u;             // wave-uniform
            uint32_t exceed = 0u;
            uint32_t m = 0u, mn = 0u, mxm = 0u, d9 = 0u;            // mxm = mx - m; d9 = the second draw's mode
            double c0 = 0.0;                                         // A - B d9 (squeeze, below); usable iff -B <= c0 < 0
            const double Bd = (double)pop2 + 2.0;
            long long gx = 0;                                        // G of the two cells that depend on x0 only
            double d6 = 0.0, d10 = 0.0, d11 = 0.0;
            // 256 more first draws: lane l makes call filled / 4 + l and writes X[4 c .. 4 c + 3] as offsets into the
            // window (u16), one 8-byte store.  The ring holds X[next .. filled): at most 63 + 256 of its 512 entries
            auto first_draw = [&](uint32_t u) {
                uint32_t lo = tab_guide[u >> GUIDE_SH], hi = tab_guide[(u >> GUIDE_SH) + 1u];
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (u < tab_thr[mid]) hi = mid; else lo = mid + 1u;
                }
                return lo;
            };
            auto refill = [&]() {
                const uint32_t c = (filled >> 2) + lane;
                const U4 o = philox4x32_10(c, ci, cj, TAG_LSX, k0, k1);
                const uint32_t e0 = first_draw(o.x), e1 = first_draw(o.y), e2 = first_draw(o.z), e3 = first_draw(o.w);
                uint2 v; v.x = e0 | (e1 << 16); v.y = e2 | (e3 << 16);
                *reinterpret_cast<uint2*>(&x_ring[(4u * c) & (XRING - 1u)]) = v;
                filled += 256u;
                __syncthreads();                         // (one wave per workgroup: orders the stores before the reads)
            };
            // HRUA set-up of the second draw for first result x0 (the expressions of the general path)
            // the window entries [c_lo, c_lo + c_n) around the first draw's mode have their look-ups in LDS, behind the thresholds
            const uint32_t c_word0 = (tab_n + 3u) & ~3u;
            uint32_t c_lo = 0u, c_n = 0u;
            if (LGMI_LS_CACHE) {
                const uint32_t room = (FIRST_MAX - c_word0) / 4u;
                const float per_sd = (float)n_shuffles / (2.5f * tab_sd);
                const uint32_t h = (uint32_t)(tab_sd * __fsqrt_rn(2.0f * __logf(per_sd > 1.0f ? per_sd : 1.0f))) + 1u;
                const uint32_t me = tab_mode - tab_klo;
                const uint32_t lo = me > h ? me - h : 0u, hi = me + h + 1u < tab_n ? me + h + 1u : tab_n;
                c_n = hi - lo < room ? hi - lo : room;
                c_lo = lo + (hi - lo - c_n) / 2u;
            }
            ulonglong2* const cache = reinterpret_cast<ulonglong2*>(tab_thr + c_word0);
            auto setup = [&](uint32_t e, bool fill) {
                const uint32_t x0 = tab_klo + e;
                uint32_t good, sample;
                if (nr == 3) { good = R1; sample = C0 - x0; }
                else { good = R0 - x0; sample = C1; hr_base(rc2, good, hb); }
                const uint32_t bad = pop2 - good;
                m = sample < pop2 - sample ? sample : pop2 - sample;
                mn = good < bad ? good : bad;
                mxm = (good < bad ? bad : good) - m;
                d6 = (double)m * hb.d4 + 0.5;
                const double cap = (double)((m < mn ? m : mn) + 1u);
                const double lim = floor(d6 + lim16);
                d11 = cap < lim ? cap : lim;
                d9 = (uint32_t)floor((double)(m + 1u) * hb.c9);
                c0 = ((double)mn * (double)m - (double)mxm - 1.0) - Bd * (double)d9;    // integers below 2^53: exact
                if (!fill && e - c_lo < c_n) {
                    const ulonglong2 v = cache[e - c_lo];
                    d10 = __longlong_as_double((long long)v.x);
                    gx = (long long)v.y;
                } else {
                    const long long ga = G.ls(x0), gb = G.ls((nr == 3 ? R0 : C0) - x0);
                    const double a3 = LF.ls(mxm + d9), a2 = LF.ls(m - d9), a1 = LF.ls(mn - d9), a0 = LF.ls(d9);
                    d10 = a0 + a1 + a2 + a3;                     // the specification's order
                    gx = ga + gb;
                }
            };
            for (uint32_t e = c_lo + lane; e < c_lo + c_n; e += 64u) {
                setup(e, true);
                ulonglong2 v; v.x = (unsigned long long)__double_as_longlong(d10); v.y = (unsigned long long)gx;
                cache[e - c_lo] = v;
            }
            // (the barrier of the first refill() below orders these stores before any read)
            // one trip: candidate (wx, wy) of every lane; true when the row is finished (wave-uniform)
            auto trip = [&](uint32_t wx, uint32_t wy) {
                if (filled - next < 64u) refill();
                const double u = (double)wx + 0.5;
                const double v = (double)(int)(wy ^ 0x80000000u) + 0.5;
                const double w = d6 + d8 * v / u;
                bool acc = false;
                // the candidate's four arguments zc, mn - zc, m - zc, mx - m + zc (minority kind in the smaller part,
                // ...) ARE the table's four cells that depend on the second draw, in some order: the accepted table's
                // statistic is the G of the same four indices plus the two cells x0 alone fixes (looked up at set-up) —
                // no mapping back to "z of the first row", no second set of index arithmetic
                uint32_t zc = 0u;
                if (!(w < 0.0 || w >= d11)) {
                    zc = (uint32_t)floor(w);
                    // ---- squeeze (round 3): the test 2 ln x <= t(zc), t = ln f(zc) / f(d9), decided from closed-form
                    // bounds of t for all but a few per cent of the candidates, so that their four log-factorial look-ups
                    // are never made (the kernel is bound by the L1's rate of scattered look-ups).  With
                    // r(k) = f(k+1)/f(k) = (mn-k)(m-k) / D_k, D_k = (k+1)(mx-m+k+1), one has r(k) - 1 = (A - B k) / D_k,
                    // A = mn m - (mx-m) - 1, B = pop + 2, and d9 = floor(A/B) + 1, i.e. c0 = A - B d9 in [-B, 0): every
                    // term of t = sum ln r(k) has the same sign on either side of d9.  ln(1+s) <= s and
                    // ln(1+s) >= s / (1+s), D_k increasing and r(k) decreasing in k give, for zc = d9 + d (d >= 1),
                    //     S1 D_{zc-1} / (D_{d9} num_r(zc-1)) <= t <= S1 / D_{zc-1},   S1 = d c0 - B d(d-1)/2 <= 0,
                    // and for zc = d9 - d,   -S2 / D_zc <= t <= -S2 D_zc / (D_{d9-1} num_r(zc)),   S2 = d c0 + B d(d+1)/2 >= 0
                    // (num_r(k) = (mn-k)(m-k)).  Compared by cross-multiplication (every denominator is positive): no
                    // division.  2 ln x comes from the hardware's f32 log, widened by 2e-4 either way: a candidate is only
                    // decided here when it is that far from both bounds, so the decision is the exact test's — which the
                    // undecided ones still take.  Nothing changes in the specification or in any result.
                    // Both sides of the mode in one branch-free form: with dd = zc - d9, k = zc - [dd > 0], j = d9 - [dd < 0],
                    // S = dd c0 - B dd (dd - 1) / 2 <= 0, P = D_j num_r(k):
                    //     dd > 0:  S D_k / P <= t <= S / D_k          dd < 0:  S / D_k <= t <= S D_k / P
                    // (dd = 0: t = 0, the same formulas with S = 0).  The first version had one branch per side of the mode and
                    // spent more on exec-mask bookkeeping than on arithmetic: it paid only where the L1 bound the loop (a
                    // row-level switch took it at >= 65,536 common reads; north-star 245 -> 233 ms).  Branch-free it is
                    // 65 instructions shorter per trip and pays everywhere: north-star 233 -> 215 ms, cfg5 1,260 -> 1,115,
                    // cfg3 145 -> 145, cfg2 7.7 -> 7.5 — the switch is gone.
                    bool decided = false;
                    if (c0 < 0.0 && c0 >= -Bd) {
                        // x = u 2^-32; v_log_f32 is log2 to ~1 ulp (|log2| <= 32: 4e-6), far inside the 2e-4 margin
                        const double lx = (double)((__builtin_amdgcn_logf((float)u) - 32.0f) * 1.3862943611198906f);
                        const double lx_hi = lx + 2e-4, lx_lo = lx - 2e-4;
                        const int dd = (int)zc - (int)d9;
                        const bool up = dd > 0;
                        const double ddf = (double)dd;
                        const double S = ddf * c0 - Bd * (0.5 * ddf * (ddf - 1.0));
                        const double kf = (double)zc - (up ? 1.0 : 0.0), jf = (double)d9 - (dd < 0 ? 1.0 : 0.0), dm = (double)mxm;
                        const double Dk = (kf + 1.0) * (dm + kf + 1.0);
                        const double P = ((jf + 1.0) * (dm + jf + 1.0)) * (((double)mn - kf) * ((double)m - kf));
                        const bool yes = lx_hi * (up ? P : Dk) <= (up ? S * Dk : S);
                        const bool no = lx_lo * (up ? Dk : P) > (up ? S : S * Dk);
                        acc = yes;
                        decided = yes | no;
                    }
                    if (!decided) {
                        const double l3 = LF.ls(mxm + zc), l2 = LF.ls(m - zc), l1 = LF.ls(mn - zc), l0 = LF.ls(zc);
                        const double tt = d10 - (l0 + l1 + l2 + l3);
                        const double x = u * 2.3283064365386963e-10;
                        acc = le_exp(x * x, tt);                     // 2 ln x <= tt
                    }
                }
                const unsigned long long bal = __ballot(acc);
                if (acc) {
                    const long long g3 = G.ls(mxm + zc), g2 = G.ls(m - zc), g1 = G.ls(mn - zc), g0 = G.ls(zc);
                    const long long gx_this = gx;                    // (before the set-up below overwrites it)
                    const uint32_t rank = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    setup(x_ring[(next + rank) & (XRING - 1u)], false);
                    const long long ss = g3 + g2 + g1 + g0 + gx_this;   // integers: any order
                    // tables are numbered in lane order; those beyond n_shuffles are surplus of the last trip
                    exceed += (uint32_t)((ss >= sobs) & (total + rank < n_shuffles));
                }
                const uint32_t na = (uint32_t)__popcll(bal);
                next += na;
                total += na;
                return total >= n_shuffles;
            };
            refill();
            setup(x_ring[lane], false);
            next = 64u;
            for (uint32_t c = 0u;; ++c) {                        // the row ends for all lanes together
                const U4 o = philox4x32_10(c, ci, cj, TAG_LSC + lane, k0, k1);
                if (trip(o.x, o.y)) break;
                if (trip(o.z, o.w)) break;
            }
            __syncthreads();                                     // the ring's last reads precede the next row's stores
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) exceed += __shfl_xor(exceed, o);
            if (lane == 0) {
                out_exceed[r] = exceed;
                if (out_p) out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
            }
            continue;
        }

        GState g;
        g.s = lane; g.phase = 3; g.call = 0; g.d = 0; g.ss = 0;
        g.rr0 = 0; g.rr1 = 0; g.pop_all = 0; g.cc = 0; g.pop = 0; g.xa = 0; g.sp1 = 0; g.sp2 = 0; g.sp3 = 0; g.spare = 0; g.aft = 0;
        g.good = 0; g.sample = 0; g.m = 0; g.mn = 0; g.mx = 0; g.d6 = 0; g.d8 = 0; g.d10 = 0; g.d11 = 0;
        g.rem_total = 0; g.rem_good = 0; g.left = 0;
        uint32_t rr2 = 0;
        uint32_t exceed = 0;
        bool need_begin = g.s < n_shuffles;   // lane has a shuffle to start
        bool have_z = false;
        uint32_t z = 0;
        // the first real draw of a shuffle when the row has a threshold table: one 32-bit word, inverse CDF
        auto table_draw = [&]() {
            const U4 o = philox4x32_10(g.s, ci, cj, TAG_PERMGEN + g.call, k0, k1);
            g.call++;
            g.sp1 = o.y; g.sp2 = o.z; g.sp3 = o.w; g.spare = 2; g.aft = 1;   // the next two pairs of uniforms come from this call
            // smallest e with u < thr[e] (the last entry when there is none); the guide table narrows the
            // search to the entries between the answers for the bucket's first value and the next bucket's
            uint32_t lo = tab_guide[o.x >> GUIDE_SH], hi = tab_guide[(o.x >> GUIDE_SH) + 1u];
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (o.x < tab_thr[mid]) hi = mid; else lo = mid + 1u;
            }
            return tab_klo + lo;
        };
        if (need_begin) {
            g.rr0 = R0; g.rr1 = R1; rr2 = R2; g.pop_all = N; g.ss = 0; g.d = 0; g.call = 0; g.spare = 0; g.aft = 0;
            g.cc = C0; g.pop = N;
            if (tab_ok) { z = table_draw(); have_z = true; need_begin = false; }
        }

        for (;;) {
            // ---- (1) one candidate / urn step for the lanes that are inside a draw: the next pair of 24-bit
            //          uniforms, two pairs per Philox call (pair A = words 1, 2 >> 8, pair B = word 3 >> 8 and the low
            //          bytes of words 1 - 3; word 0 is the threshold-table draw's)
            const bool in_draw = !need_begin && g.phase != 3;
            double ux = 0.5, uy = 0.5;
            if (in_draw) {
                if (g.spare == 0) {
                    const U4 o = philox4x32_10(g.s, ci, cj, TAG_PERMGEN + g.call, k0, k1);
                    g.call++;
                    g.sp1 = o.y; g.sp2 = o.z; g.sp3 = o.w; g.spare = 2;
                }
                uint32_t vx, vy;
                if (g.spare == 2) { vx = g.sp1 >> 8; vy = g.sp2 >> 8; }
                else { vx = g.sp3 >> 8; vy = ((g.sp1 & 0xFFu) << 16) | ((g.sp2 & 0xFFu) << 8) | (g.sp3 & 0xFFu); }
                g.spare--;
                g.aft = 0;
                ux = ((double)vx + 0.5) * 5.9604644775390625e-08;
                uy = ((double)vy + 0.5) * 5.9604644775390625e-08;
            }
            if (in_draw && g.phase == 1) {
                const double x = ux, y = uy;
                const double w = g.d6 + g.d8 * (y - 0.5) / x;
                if (!(w < 0.0 || w >= g.d11)) {
                    const uint32_t zc = (uint32_t)floor(w);
                    const double tt = g.d10 - (LF[zc] + LF[g.mn - zc] + LF[g.m - zc] + LF[g.mx - g.m + zc]);
                    const bool acc = le_exp(x * x, tt);   // 2 ln x <= tt
                    if (acc) {
                        z = zc;
                        if (g.good > g.pop - g.good) z = g.m - z;   // z counted the minority kind
                        if (g.m < g.sample) z = g.good - z;         // drew the complement
                        have_z = true;
                    }
                }
            } else if (in_draw && g.phase == 2) {
                // the WHOLE urn draw inside this trip (at most 9 steps, one pair of the shuffle's stream per step, in
                // order): with one step per trip a small draw cost as many trips of the state machine as it has steps,
                // and rows of a few hundred reads — the footprint-shaped regime — are mostly such draws
                for (;;) {
                    if (g.left > 0u && g.rem_good > 0u && g.rem_total > g.rem_good) {
                        if ((uint32_t)(ux * (double)g.rem_total) < g.rem_good) g.rem_good--;
                        g.rem_total--;
                        g.left--;
                    }
                    if (!(g.left > 0u && g.rem_good > 0u && g.rem_total > g.rem_good)) break;
                    if (g.spare == 0) {
                        const U4 o = philox4x32_10(g.s, ci, cj, TAG_PERMGEN + g.call, k0, k1);
                        g.call++;
                        g.sp1 = o.y; g.sp2 = o.z; g.sp3 = o.w; g.spare = 2;
                    }
                    const uint32_t vx = g.spare == 2 ? g.sp1 >> 8 : g.sp3 >> 8;
                    g.spare--;
                    ux = ((double)vx + 0.5) * 5.9604644775390625e-08;
                }
                if (g.rem_total == g.rem_good) g.rem_good -= g.left;
                z = g.good - g.rem_good;
                if (g.m < g.sample) z = g.good - z;
                have_z = true;
            }
            // ---- (2) lanes whose draw just finished (or that start a shuffle): book the result, close the
            //          column / the table when it is complete, set up the next real draw.
            //          g.d = 2 * column + row (row 1 only exists when three rows are non-empty)
            if (have_z || need_begin) {
                bool done = false;                                   // lane has run out of shuffles
                for (;;) {
                    // (2a) booking.  A lane that finishes a shuffle and has a table draws the next shuffle's
                    //      first result right here and books it in a second pass, so that every lane reaches
                    //      the set-up code below once, together (the table draw is never trivially determined:
                    //      the table only exists when both margins of the draw are inside (0, N))
                    while (have_z) {
                        have_z = false;
                        bool column_done = false;
                        uint32_t x0 = 0, x1 = 0, x2 = 0;
                        if ((g.d & 1) == 0) {
                            g.xa = z;
                            if (nr == 3) { g.pop -= g.rr0; g.cc -= z; g.d++; }
                            else { x0 = z; x1 = g.cc - z; column_done = true; }
                        } else {
                            x0 = g.xa; x1 = z; x2 = g.cc - z;       // the last row takes the rest of the column
                            column_done = true;
                        }
                        if (column_done) {
                            g.ss += G[x0] + G[x1] + G[x2];
                            g.rr0 -= x0; g.rr1 -= x1; rr2 -= x2;
                            g.pop_all -= ((g.d >> 1) == 0 ? C0 : C1);
                            g.d = (g.d | 1) + 1;                     // first row of the next column
                            if ((g.d >> 1) == nc - 1) {              // the last column takes what is left
                                g.ss += G[g.rr0] + G[g.rr1] + G[rr2];
                                exceed += (g.ss >= sobs);
                                g.s = atomicAdd(&next_s, 1u);
                                if (g.s >= n_shuffles) { g.phase = 3; done = true; break; }
                                g.rr0 = R0; g.rr1 = R1; rr2 = R2; g.pop_all = N; g.ss = 0; g.d = 0; g.call = 0; g.spare = 0; g.aft = 0;
                                if (tab_ok) { g.cc = C0; g.pop = N; z = table_draw(); have_z = true; }
                            }
                        }
                    }
                    if (done) break;
                    need_begin = false;
                    // (2b) parameters of draw g.d
                    if ((g.d & 1) == 0) { g.cc = ((g.d >> 1) == 0 ? C0 : C1); g.pop = g.pop_all; g.good = g.rr0; }
                    else { g.good = g.rr1; }
                    g.sample = g.cc;
                    const uint32_t pop = g.pop, good = g.good, sample = g.sample, bad = pop - good;
                    if (sample == 0u || good == 0u) { z = 0u; have_z = true; continue; }
                    if (bad == 0u) { z = sample; have_z = true; continue; }
                    if (sample == pop) { z = good; have_z = true; continue; }
                    g.m = sample < pop - sample ? sample : pop - sample;
                    if (g.m < 10u) {
                        g.rem_total = pop; g.rem_good = good; g.left = g.m;
                        g.phase = 2;
                    } else {
                        // (with the reciprocals at hand the three quotients are five flops: no cache per draw position)
                        if (rcp.pop != pop) hr_recip(pop, rcp);
                        HrBase hb;
                        hr_base(rcp, good, hb);
                        g.mn = good < bad ? good : bad;
                        g.mx = good < bad ? bad : good;
                        g.d6 = (double)g.m * hb.d4 + 0.5;
                        // the draw right after the table draw of a 3 x 2 / 2 x 3 row takes the row's hat width
                        const bool bounded = d7max > 0.0 && g.aft && pop == key_pop2 && (nr == 3 ? good == key2 : sample == key2);
                        const double d7 = bounded ? d7max : det_sqrt((double)(pop - g.m) * (double)g.m * hb.cvar + 0.5);
                        const uint32_t d9 = (uint32_t)floor((double)(g.m + 1u) * hb.c9);
                        g.d10 = LF[d9] + LF[g.mn - d9] + LF[g.m - d9] + LF[g.mx - g.m + d9];
                        g.d8 = HRUA_D1 * d7 + HRUA_D2;
                        const double cap = (double)((g.m < g.mn ? g.m : g.mn) + 1u);
                        const double lim = floor(g.d6 + 16.0 * d7);
                        g.d11 = cap < lim ? cap : lim;
                        g.phase = 1;
                    }
                    break;
                }
            }
            if (!__any(g.phase != 3)) break;   // wave-uniform exit: every lane has finished its shuffles
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) exceed += __shfl_xor(exceed, o);
        if (lane == 0) {
            out_exceed[r] = exceed;
            if (out_p) out_p[r] = (1.0 + (double)exceed) / ((double)n_shuffles + 1.0);
        }
    }   // rows of the batch
    }   // batch loop
}

// self-test hook (lgmi_selftest_le_exp): the decision of le_exp, the decision of det_exp alone, and both exponentials
__global__ void k_selftest_le_exp(uint64_t n, const double* __restrict__ x2, const double* __restrict__ t,
                                  uint8_t* __restrict__ fast, uint8_t* __restrict__ det, double* __restrict__ e_hw,
                                  double* __restrict__ e_det)
{
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    fast[k] = le_exp(x2[k], t[k]) ? 1 : 0;
    const double d = det_exp(t[k]);
    det[k] = x2[k] <= d ? 1 : 0;
    e_hw[k] = (double)__expf((float)t[k]);
    e_det[k] = d;
}
void launch_selftest_le_exp(hipStream_t st, uint64_t n, const double* x2, const double* t, uint8_t* fast, uint8_t* det,
                            double* e_hw, double* e_det)
{
    if (!n) return;
    hipLaunchKernelGGL(k_selftest_le_exp, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, x2, t, fast, det, e_hw, e_det);
}

// LGMI_PERM_ENUM_MAX: largest number of candidate tables a larger-than-2x2 row is enumerated at (0: never; tests compare
// both paths with the CPU specification, which has the same switch)
static uint32_t perm_enum_max()
{
    static const uint32_t v = [] {
        const char* e = getenv("LGMI_PERM_ENUM_MAX");
        const long x = e ? atol(e) : 4096;
        return (uint32_t)(x < 0 ? 0 : x > 8192 ? 8192 : x);
    }();
    return v;
}

// LGMI_PERM_SIX_PTS: sixteenths of a chord per shuffle a six-cell row may cost on the exact path (0: every such row is sampled;
// the CPU specification has the same switch)
static uint32_t perm_six_pts()
{
    static const uint32_t v = [] {
        const char* e = getenv("LGMI_PERM_SIX_PTS");
        const long x = e ? atol(e) : 16;
        return (uint32_t)(x < 0 ? 0 : x > 65536 ? 65536 : x);
    }();
    return v;
}

void launch_perm_fast(hipStream_t st, const PermArgs& a)
{
    if (!a.max_rows) return;
    const uint64_t chunks = (a.max_rows + 63) / 64;
    PermArgs b = a;
    b.enum_max = perm_enum_max();
    hipLaunchKernelGGL(k_perm_fast, dim3((uint32_t)std::min<uint64_t>(chunks, 256ull * 32ull)), dim3(64), 0, st, b);
}

void launch_perm_general(hipStream_t st, const PermArgs& a, hipEvent_t after_exact)
{
    if (!a.max_rows || (!a.n_shuffles && !a.exact_2x2)) { if (after_exact) (void)hipEventRecord(after_exact, st); return; }
    // a fixed grid (16 one-wave workgroups per CU, 8 KB of LDS each) whose waves
    // take the queued rows from a shared counter
    // (LGMI_PERM_WPC: one-wave workgroups per CU, for occupancy experiments — tools/abl_perm.sh; 16 = four per SIMD)
    // as many as the LDS of a CU holds, 16 at most (12 or 8 run the loop as fast as 16: profiles/r03_perm_general_occupancy.txt)
    static const int wpc = [] {
        const int fit = (int)((160u * 1024u) / (FIRST_MAX * 4u + (GUIDE_N + 1u) * 2u + XRING * 2u + 64u));
        const int dflt = fit < 16 ? (fit < 1 ? 1 : fit) : 16;
        const char* e = getenv("LGMI_PERM_WPC");
        const int v = e ? atoi(e) : dflt;
        return v >= 1 && v <= 32 ? v : dflt;
    }();
    PermArgs b = a;
    b.enum_max = perm_enum_max();
    // LGMI_PERM_NO_SECOND_LIST=1 (tests): as if the queue filled more than half of its buffer — k_perm_enum marks the rows it
    // finishes in the queue and k_perm_general skips them, instead of walking a second list
    static const bool no_second = [] { const char* e = getenv("LGMI_PERM_NO_SECOND_LIST"); return e && atoi(e) != 0; }();
    if (no_second) b.max_rows = 1;
    b.six_pts = perm_six_pts();
    if (b.enum_max) hipLaunchKernelGGL(k_perm_enum, dim3(256 * 16), dim3(64), 0, st, b);
    if (b.six_pts) hipLaunchKernelGGL(k_perm_six, dim3(256 * 4 * LGMI_SIX_WPS), dim3(64), 0, st, b);
    if (after_exact) (void)hipEventRecord(after_exact, st);      // enumeration + perimeter walk done: what follows is sampling
    if (a.n_shuffles) hipLaunchKernelGGL(k_perm_general, dim3(256 * wpc), dim3(64), 0, st, b);   // (exact-p mode without shuffles: nothing to sample)
}

}  // namespace lgmi
