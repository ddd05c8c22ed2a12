/*
 * lgmi_perm_oracle.c — CPU specification of the permutation-test p-value of a site
 * pair.  TEST INFRASTRUCTURE ONLY (see lgmi_oracle.c).
 *
 * PARITY UNPINNED: the reference (gxiaolab/L-GIREMI v0.2.4) has no permutation test
 * (its only p-value is the ECDF `mip`, src/giremi/stat.py:7-29); BASELINE.json's
 * north_star asks for one.  This file IS the specification; the HIP
 * kernel (l-giremi_amd/csrc/perm.hip) must reproduce `exceed` bit for bit.
 *
 * Null hypothesis: the class labels of site j are shuffled uniformly among the N
 * reads common to both sites; the 3x3 table is then multivariate hypergeometric
 * with the observed margins.  Statistic: MI, compared through
 *     S(T) = sum_ab G[T_ab],  G[n] = round(n*ln(n) * 2^28)   (MI*N = S/2^28 - const for fixed margins)
 * in 64-bit INTEGER arithmetic: the sum does not depend on the order of the cells, so
 * tables made of the same cell counts (very common with small counts) tie exactly.
 * p = (1 + #{S(T_s) >= S(T_obs)}) / (n_shuffles + 1).
 *
 *  - <= 1 non-empty row or column: every shuffle gives the same table, exceed = n_shuffles.
 *  - exactly 2 x 2 non-empty: one degree of freedom k.  Shuffle s draws k by inverse
 *    CDF with the "as or more extreme" set enumerated first, so it lands in that set
 *    iff u_s < P_tail; P_tail is evaluated exactly from log-factorials (no table is
 *    materialised).  u_s are 32-bit Philox outputs, threshold floor(P_tail * 2^32).
 *  - anything larger: shuffle s draws the table cell by cell with conditional
 *    hypergeometric draws (simple urn scheme for small samples, Stadlober's HRUA
 *    ratio-of-uniforms otherwise) and evaluates S.
 *
 * Everything that decides an outcome uses only +,-,*,/ and comparisons on doubles
 * (compiled with -ffp-contract=off), integer arithmetic, and the tables G[] (fixed
 * point) and LF[] (ln n!, double) computed once with libm on the host — so the CPU and the GPU agree
 * bit for bit.  exp, log and sqrt are the deterministic routines below, not libm.
 *
 * Published algorithms restated here: Philox4x32-10 (Salmon, Moraes, Dror, Shaw,
 * SC'11); HRUA (Stadlober, "The ratio of uniforms approach for generating discrete
 * random variates", J. Comput. Appl. Math. 31 (1990)).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define TAG_PERM2X2 0x5eed0004u
#define TAG_PERMGEN 0x60000000u

/* ------------------------------------------------------------------ Philox4x32-10 */
static void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    int r;
    for (r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ------------------------------------------------------------------ deterministic exp / log / sqrt */
static double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
static const double INV_LN2 = 1.44269504088896338700e+00;
/* 1/n, correctly rounded (IEEE division of exact constants, folded by the compiler) */
static const double INV_N[24] = {0.0, 1.0, 1.0 / 2.0, 1.0 / 3.0, 1.0 / 4.0, 1.0 / 5.0, 1.0 / 6.0, 1.0 / 7.0, 1.0 / 8.0,
                                 1.0 / 9.0, 1.0 / 10.0, 1.0 / 11.0, 1.0 / 12.0, 1.0 / 13.0, 1.0 / 14.0, 1.0 / 15.0,
                                 1.0 / 16.0, 1.0 / 17.0, 1.0 / 18.0, 1.0 / 19.0, 1.0 / 20.0, 1.0 / 21.0, 1.0 / 22.0,
                                 1.0 / 23.0};

/* exp(x) for x <= 0, relative error < 1e-12 (all that a 2^-32 threshold needs), made of fma(),
 * floor() and exact scalings only: the same bits on the CPU and on the GPU */
double lgo_det_exp(double x)
{
    double k, r, p;
    int ki;
    if (!(x > -745.0)) return 0.0;
    if (x > 0.0) x = 0.0;
    k = floor(fma(x, INV_LN2, 0.5));
    r = fma(-k, LN2_HI, x);
    r = fma(-k, LN2_LO, r);                       /* |r| <= 0.347 */
    p = 2.755731922398589e-07;                    /* 1/10! ... 1/2!, Horner */
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
    ki = (int)k;
    if (ki >= -1000) return p * bits_to_double((uint64_t)(ki + 1023) << 52);
    return (p * bits_to_double((uint64_t)(-1000 + 1023) << 52)) * bits_to_double((uint64_t)(ki + 1000 + 1023) << 52);
}

double lgo_det_log(double x)   /* x > 0, normal */
{
    uint64_t b = double_to_bits(x);
    int e = (int)((b >> 52) & 0x7FF) - 1023, n;
    double m = bits_to_double((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);   /* [1,2) */
    double f, f2, s;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    f = (m - 1.0) / (m + 1.0);
    f2 = f * f;
    s = 0.0;
    for (n = 23; n >= 1; n -= 2) s = s * f2 + INV_N[n];   /* sum f^(2j)/(2j+1) */
    return (double)e * LN2_HI + (2.0 * f * s + (double)e * LN2_LO);
}

double lgo_det_sqrt(double a)   /* a > 0: division-free Newton on y = 1/sqrt(a) from a bit-level guess, then a*y */
{
    const uint64_t b = double_to_bits(a);
    double y = bits_to_double(0x5FE6EB50C7B537A9ull - (b >> 1));
    int n;
    for (n = 0; n < 4; ++n) {
        const double t = a * y;
        const double h = fma(-t, y, 1.0);          /* 1 - a y^2 */
        y = fma(y * 0.5, h, y);
    }
    return a * y;
}

/* ------------------------------------------------------------------ tables */
typedef struct { int64_t* G; double* LF; uint32_t len; } perm_tables;

static int tables_init(perm_tables* t, uint32_t max_n)
{
    uint32_t n;
    t->len = max_n + 1;
    t->G = (int64_t*)malloc((size_t)t->len * sizeof(int64_t));
    t->LF = (double*)malloc((size_t)t->len * sizeof(double));
    if (!t->G || !t->LF) { free(t->G); free(t->LF); return -1; }
    t->G[0] = 0;
    for (n = 1; n <= max_n; ++n) t->G[n] = llrint((double)n * log((double)n) * 268435456.0);   /* 2^28 */
    for (n = 0; n <= max_n; ++n) t->LF[n] = lgamma((double)n + 1.0);
    return 0;
}

/* ------------------------------------------------------------------ uniform streams */
typedef struct {
    uint32_t c0, c1, c2, k0, k1, call;
    uint32_t buf[4];
    int have;          /* pairs of uniforms still unused in buf: 0, 1 or 2 */
    int after_table;   /* the last draw was the threshold-table draw (hg_draw: the row-constant hat width) */
} gen_stream;

/* Uniforms of the conditional draws: 24-bit, u = (v + 0.5) * 2^-24 (exact in double, never 0 or 1), TWO pairs per
 * Philox call (round 3; rounds 1-2 took one 32-bit pair per call and threw half of every call away):
 *     pair A = (word1 >> 8, word2 >> 8)
 *     pair B = (word3 >> 8, (word1 & 0xFF) << 16 | (word2 & 0xFF) << 8 | (word3 & 0xFF))
 * i.e. words 1-3 are cut into four disjoint 24-bit fields.  Word 0 belongs to the threshold-table draw: call 0 of a
 * shuffle whose first draw has a table spends it there and its pairs A, B serve the draws that follow; in every other
 * call word 0 is not used.  A rejection loop consumes A, B, then the next call's A, B, ...: the kernel evaluates both
 * candidates of a call side by side (two independent dependency chains per lane) and takes the first acceptable one,
 * so one call ends a ratio-of-uniforms draw with probability 1 - 0.28^2 = 0.92 instead of 0.72.  24 bits: the
 * candidate w = d6 + d8 (y - 1/2) / x moves by <= 1e-5 of an integer bin per step of y, the acceptance test by 6e-8
 * relative per step of x — the same discretisation single-precision generators live with. */
static void next_pair(gen_stream* g, double* u0, double* u1)
{
    uint32_t vx, vy;
    if (!g->have) {
        philox(g->c0, g->c1, g->c2, TAG_PERMGEN + g->call, g->k0, g->k1, g->buf);
        g->call++;
        g->have = 2;
    }
    if (g->have == 2) {
        vx = g->buf[1] >> 8;
        vy = g->buf[2] >> 8;
    } else {
        vx = g->buf[3] >> 8;
        vy = ((g->buf[1] & 0xFFu) << 16) | ((g->buf[2] & 0xFFu) << 8) | (g->buf[3] & 0xFFu);
    }
    g->have--;
    g->after_table = 0;
    *u0 = ((double)vx + 0.5) * 5.9604644775390625e-08;     /* 2^-24 */
    *u1 = ((double)vy + 0.5) * 5.9604644775390625e-08;
}

/* ------------------------------------------------------------------ hypergeometric draw */
static const double HRUA_D1 = 1.7155277699214135;   /* 2*sqrt(2/e)     */
static const double HRUA_D2 = 0.8989161620588988;   /* 3 - 2*sqrt(3/e) */

/* The first draw of a shuffle that is not trivially determined has the same parameters in every shuffle
 * of a row.  When its distribution fits a window of at most FIRST_MAX values (mode +- (6.5 sd + 4)) it is
 * drawn by inverse CDF instead of rejection: one 32-bit Philox word u, result = the smallest k of the window
 * with u < thr[k - klo] (the last k of the window when there is none: the window misses < 1e-10 of the mass).
 * The thresholds are integer prefix sums, so they do not depend on the order of the additions:
 *     q[e]   = rint(pmf(klo + e) * 2^52)                                 (u64; pmf from the log-factorials at the
 *              first entry of a segment, then by the ratio pmf(k) = pmf(k-1) (good-k+1)(sample-k+1)/(k (pop-good-sample+k)))
 *     the window is cut into 64 contiguous segments of seg = ceil(n / 64) entries (one per GPU lane);
 *     thr[e] = min(2^32 - 1, (sum of q over the entries of e's segment up to e) >> 20
 *                            + (sum of q over all earlier segments) >> 20)
 * i.e. floor(2^32 * CDF) to within 2 units (the two shifts are taken separately so that a lane only has to
 * keep 32 bits per entry). */
#define FIRST_MAX 2032
typedef struct {
    int valid;
    uint32_t pop, good, sample, klo, n;
    /* 3 x 2 / 2 x 3 rows: the HRUA draw that follows the table draw — population pop2 and, with nr3, good == key2,
     * else sample == key2 — takes d7_next (> 0) instead of its own sqrt(var + 0.5): see hrua_width_bound() */
    uint32_t pop2, key2;
    int nr3;
    double d7_next;
    uint32_t thr[FIRST_MAX];
} first_table;

static void first_table_build(const perm_tables* t, uint32_t pop, uint32_t good, uint32_t sample, first_table* ft)
{
    const uint32_t m = sample < pop - sample ? sample : pop - sample;
    const uint32_t kmin = sample + good > pop ? sample + good - pop : 0, kmax = good < sample ? good : sample;
    double c0, var, sd, p;
    uint32_t mode, w, khi, seg, e, l;
    uint64_t earlier = 0;
    ft->valid = 0;
    ft->d7_next = 0.0; ft->pop2 = 0; ft->key2 = 0; ft->nr3 = 0;
    if (sample == 0 || good == 0 || good == pop || sample == pop || m < 10) return;
    p = (double)good / (double)pop;
    var = (double)sample * p * (1.0 - p) * (double)(pop - sample) / (double)(pop - 1);
    sd = lgo_det_sqrt(var + 1.0);
    w = (uint32_t)floor(6.5 * sd) + 4u;
    mode = (uint32_t)(((uint64_t)(sample + 1) * (uint64_t)(good + 1)) / ((uint64_t)pop + 2));
    if (mode < kmin) mode = kmin;
    if (mode > kmax) mode = kmax;
    ft->klo = mode - kmin > w ? mode - w : kmin;
    khi = kmax - mode > w ? mode + w : kmax;
    ft->n = khi - ft->klo + 1;
    if (ft->n > FIRST_MAX) return;
    c0 = t->LF[good];
    c0 += t->LF[pop - good];
    c0 += t->LF[sample];
    c0 += t->LF[pop - sample];
    c0 -= t->LF[pop];
    seg = (ft->n + 63u) / 64u;
    for (l = 0; l < 64; ++l) {
        uint64_t loc = 0;
        double pm = 0.0;
        for (e = l * seg; e < ft->n && e < (l + 1) * seg; ++e) {
            const uint32_t k = ft->klo + e;
            uint64_t thr;
            if (e == l * seg) {
                /* first entry of a segment: from the log-factorials */
                double x = c0;
                x -= t->LF[k];
                x -= t->LF[good - k];
                x -= t->LF[sample - k];
                x -= t->LF[pop - good - sample + k];
                pm = lgo_det_exp(x);
            } else {
                /* the others: pmf(k) = pmf(k-1) (good-k+1)(sample-k+1) / (k (pop-good-sample+k)) */
                const double num = (double)(good - k + 1u) * (double)(sample - k + 1u);
                const double den = (double)k * (double)(pop - good - sample + k);
                pm = pm * num / den;
            }
            loc += (uint64_t)rint(pm * 4503599627370496.0);
            thr = (uint64_t)(uint32_t)(loc >> 20) + (earlier >> 20);
            ft->thr[e] = thr >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)thr;
        }
        earlier += loc;
    }
    ft->pop = pop; ft->good = good; ft->sample = sample;
    ft->valid = 1;
}

/* number of "good" items in `sample` draws without replacement from pop = good + bad */
static uint32_t hg_draw(const perm_tables* t, uint32_t pop, uint32_t good, uint32_t sample, gen_stream* g,
                        const first_table* ft)
{
    const uint32_t bad = pop - good;
    const uint32_t m = sample < pop - sample ? sample : pop - sample;
    uint32_t z;
    if (sample == 0 || good == 0) return 0;
    if (bad == 0) return sample;
    if (sample == pop) return good;
    if (ft && ft->valid && g->call == 0 && ft->pop == pop && ft->good == good && ft->sample == sample) {
        /* the first real draw of the shuffle: inverse CDF on the row's table */
        uint32_t lo = 0, hi = ft->n - 1, u;
        philox(g->c0, g->c1, g->c2, TAG_PERMGEN + g->call, g->k0, g->k1, g->buf);
        g->call++;
        g->have = 2;                 /* words 1 - 3 of this call are the next two pairs of uniforms */
        g->after_table = 1;
        u = g->buf[0];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (u < ft->thr[mid]) hi = mid; else lo = mid + 1;
        }
        return ft->klo + lo;
    }
    if (m < 10) {
        /* urn scheme on the smaller of the sample and its complement */
        uint32_t rem_total = pop, rem_good = good, left = m;
        while (left > 0 && rem_good > 0 && rem_total > rem_good) {
            double u, unused;
            next_pair(g, &u, &unused);
            if ((uint32_t)(u * (double)rem_total) < rem_good) rem_good--;
            rem_total--;
            left--;
        }
        if (rem_total == rem_good) rem_good -= left;   /* only good items remain */
        z = good - rem_good;                           /* good items among the m drawn */
    } else {
        const uint32_t mn = good < bad ? good : bad, mx = good < bad ? bad : good;
        /* the three quotients depend on (pop, good) only: the kernel keeps them across shuffles */
        /* quotients by pop, pop - 1, pop + 2 as products with their reciprocals: within a row most draws share
         * the population size, so the GPU keeps the three reciprocals and never divides per draw */
        const double rp = 1.0 / (double)pop, rp1 = 1.0 / (double)(pop - 1), rp2 = 1.0 / ((double)pop + 2.0);
        const double d4 = (double)mn * rp;
        const double cvar = d4 * (1.0 - d4) * rp1;
        const double c9 = (double)(mn + 1) * rp2;
        const double d6 = (double)m * d4 + 0.5;
        const int bounded = ft && ft->valid && ft->d7_next > 0.0 && g->after_table && pop == ft->pop2 &&
                            (ft->nr3 ? good == ft->key2 : sample == ft->key2);
        const double d7 = bounded ? ft->d7_next : lgo_det_sqrt((double)(pop - m) * (double)m * cvar + 0.5);
        const double d8 = HRUA_D1 * d7 + HRUA_D2;
        const uint32_t d9 = (uint32_t)floor((double)(m + 1) * c9);   /* mode (may be off by one: harmless) */
        const double d10 = t->LF[d9] + t->LF[mn - d9] + t->LF[m - d9] + t->LF[mx - m + d9];
        const double cap = (double)((m < mn ? m : mn) + 1u);
        const double lim = floor(d6 + 16.0 * d7);
        const double d11 = cap < lim ? cap : lim;
        for (;;) {
            double x, y, w, tt;
            uint32_t zc;
            next_pair(g, &x, &y);
            w = d6 + d8 * (y - 0.5) / x;
            if (w < 0.0 || w >= d11) continue;
            zc = (uint32_t)floor(w);
            tt = d10 - (t->LF[zc] + t->LF[mn - zc] + t->LF[m - zc] + t->LF[mx - m + zc]);
            /* 2 ln x <= tt, without a log.  (Stadlober's two squeeze tests are left out: they only save the exp on a
               scalar machine; on a 64-lane wave some lane always needs it, and each test is a branch.) */
            if (x * x <= lgo_det_exp(tt)) { z = zc; break; }
        }
        if (good > bad) z = m - z;   /* z counted the minority kind */
    }
    if (m < sample) z = good - z;    /* drew the complement */
    return z;
}

/* ------------------------------------------------------------------ statistic */
static int64_t stat9(const perm_tables* t, const uint32_t T[9])
{
    int64_t s = 0;
    int k;
    for (k = 0; k < 9; ++k) s += t->G[T[k]];
    return s;
}

/* 2 x 2 core: rows a1 < a2, cols b1 < b2 non-empty; k = T[a2][b2] */
typedef struct { uint32_t N, K, n, kmin, kmax; double c0; } hg22;

static int64_t stat22(const perm_tables* t, const hg22* h, uint32_t k)
{
    int64_t s = 0;
    s += t->G[h->N - h->K - h->n + k];
    s += t->G[h->n - k];
    s += t->G[h->K - k];
    s += t->G[k];
    return s;
}

static double pmf22(const perm_tables* t, const hg22* h, uint32_t k)
{
    double e = h->c0;
    e -= t->LF[k];
    e -= t->LF[h->K - k];
    e -= t->LF[h->n - k];
    e -= t->LF[h->N - h->K - h->n + k];
    return lgo_det_exp(e);
}

/* 64-way strided summation, the order the GPU wave uses: term number q of a run goes to
 * accumulator q % 64; the 64 accumulators are combined by an xor butterfly. */
/* Exact mass of a range of k, as a 64-bit integer in units of 2^-62.  The range [k0, k0 + len) is cut into units
 * of UNIT = 64 consecutive values; a unit is summed in double precision — its first term from the log-factorials
 * (one look-up of four table lines per 64 values), the following ones through the hypergeometric ratio
 *     pmf(k+1) = pmf(k) (K-k)(n-k) / ((k+1)(N-K-n+k+1))
 * carried division-free over sub-blocks of SUB = 16 steps:  N <- N num,  Q <- Q den,  P <- fma(P, den, N), then
 * rQ = 1 / Q, sum += (t P) rQ and t <- (t N) rQ (products of 16 factors below 2^52 stay inside the double range) — and
 * truncated to the fixed-point grid.  The mass of a range is the INTEGER sum of its units, so it does not depend
 * on how units are dealt to GPU lanes or in which order they are added. */
#define UNIT 64
#define SUB 16
static uint64_t unit_mass(const perm_tables* t, const hg22* h, int64_t k0, int64_t len)
{
    uint32_t k = (uint32_t)k0;
    double term = pmf22(t, h, k), sum = term;
    int64_t rem = len - 1;
    while (rem > 0) {
        const int64_t m = rem < SUB ? rem : SUB;
        double P = 0.0, Nn = 1.0, Q = 1.0;
        double a = (double)(h->K - k), b = (double)(h->n - k), c = (double)(k + 1u), d = (double)(h->N - h->K - h->n + k + 1u);
        int64_t j;
        for (j = 0; j < m; ++j) {
            const double num = a * b, den = c * d;
            Nn = Nn * num;
            Q = Q * den;
            P = fma(P, den, Nn);
            a -= 1.0; b -= 1.0; c += 1.0; d += 1.0;
        }
        {
            const double rQ = 1.0 / Q;                   /* (round 4: one division per sub-block, two products) */
            sum += (term * P) * rQ;
            term = (term * Nn) * rQ;
        }
        k += (uint32_t)m;
        rem -= m;
    }
    return (uint64_t)(sum * 4611686018427387904.0);   /* 2^62 */
}

static uint64_t range_mass(const perm_tables* t, const hg22* h, int64_t k0, int64_t len)
{
    uint64_t s = 0;
    int64_t o;
    for (o = 0; o < len; o += UNIT) s += unit_mass(t, h, k0 + o, len - o < UNIT ? len - o : UNIT);
    return s;
}

/* trunc(2^32 P(S(k) >= S(k_obs))) under the hypergeometric null, in [0, 2^32]; *p_out gets P as a double */
static uint64_t ptail22(const perm_tables* t, const hg22* h, uint32_t kobs, double* p_out)
{
    const int64_t sobs = stat22(t, h, kobs);
    uint32_t kc = (uint32_t)(((uint64_t)h->n * (uint64_t)h->K) / (uint64_t)h->N);   /* S decreases up to kc, increases after */
    int64_t klo, khi;   /* tail = [kmin, klo] U [khi, kmax] */
    double var, clen;
    uint64_t s, thr;
    if (kc < h->kmin) kc = h->kmin;
    if (kc > h->kmax) kc = h->kmax;
    if (kobs <= kc) {
        int64_t lo = (int64_t)kc + 1, hi = (int64_t)h->kmax + 1;   /* first k in [kc+1, kmax] with S(k) >= sobs */
        klo = kobs;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (stat22(t, h, (uint32_t)mid) >= sobs) hi = mid; else lo = mid + 1;
        }
        khi = lo;
    } else {
        int64_t lo = (int64_t)h->kmin - 1, hi = (int64_t)kc;       /* last k in [kmin, kc] with S(k) >= sobs */
        khi = kobs;
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo + 1) / 2;
            if (stat22(t, h, (uint32_t)mid) >= sobs) lo = mid; else hi = mid - 1;
        }
        klo = lo;
    }
    var = (double)h->n * (double)h->K * (double)(h->N - h->K) * (double)(h->N - h->n)
          / ((double)h->N * (double)h->N * (double)(h->N > 1 ? h->N - 1 : 1));
    clen = (double)(khi - klo - 1);
    if (clen * clen <= 49.0 * var + 64.0) {
        /* few values are less extreme (within ~7 sigma): 1 - their mass */
        s = range_mass(t, h, klo + 1, khi - klo - 1);
        thr = s <= 4611686018427387904ull ? (4611686018427387904ull - s) >> 30 : 0;
        if (p_out) *p_out = 1.0 - (double)s * 2.168404344971009e-19;
    } else {
        /* both tails start >= ~3.5 sigma out.  When their first (largest) terms are below 2^-60 the whole set
         * weighs less than 2^-33: thr = 0.  Otherwise each tail is summed over 8 sigma + 16 values (what lies
         * beyond is below 2^-80 of it), clipped to the support. */
        const double f_lo = klo >= (int64_t)h->kmin ? pmf22(t, h, (uint32_t)klo) : 0.0;
        const double f_hi = khi <= (int64_t)h->kmax ? pmf22(t, h, (uint32_t)khi) : 0.0;
        s = 0;
        if (!(f_lo < 8.673617379884035e-19 && f_hi < 8.673617379884035e-19)) {
            const int64_t D = (int64_t)(8.0 * lgo_det_sqrt(var + 1.0)) + 16;
            int64_t lo_start = klo - D + 1, hi_end = khi + D - 1;
            if (lo_start < (int64_t)h->kmin) lo_start = (int64_t)h->kmin;
            if (hi_end > (int64_t)h->kmax) hi_end = (int64_t)h->kmax;
            if (klo >= lo_start) s += range_mass(t, h, lo_start, klo - lo_start + 1);
            if (hi_end >= khi) s += range_mass(t, h, khi, hi_end - khi + 1);
        }
        thr = s >> 30;
        if (thr > 4294967296ull) thr = 4294967296ull;
        if (p_out) *p_out = (double)s * 2.168404344971009e-19;
    }
    if (p_out) { if (*p_out > 1.0) *p_out = 1.0; if (*p_out < 0.0) *p_out = 0.0; }
    return thr;
}

/* ------------------------------------------------------------------ binomial draw (2 x 2 tables)
 * For a 2 x 2 table the probability that a shuffle is "as or more extreme" is known exactly (ptail22), so the
 * number of such shuffles among n_shuffles is Binomial(n_shuffles, P) and is drawn as ONE binomial variate
 * instead of n_shuffles Bernoulli trials.  P enters as thr = trunc(P * 2^32) in units of 2^-32.
 *   thr == 0 -> 0;  thr >= 2^32 -> n;  thr > 2^31: n - Binomial(n, 1 - p)  (1 - p is exact)
 *   n p < 10 : sequential inversion from 0 (BINV; restart with fresh uniforms beyond np + 10 sqrt(npq + 1)), carried on
 *              the scale of x! — U = x! (u - F(x - 1)) against T = x! f(x): U <- (U - T) x, T <- T (n - x + 1) (p / q) —
 *              so that a step has no division (round 4; 44! = 2.7e54 is the largest factor)
 *   else     : Hoermann's transformed rejection BTRS (1993), the acceptance test taken against the exact
 *              log-factorial table: v alpha / (a / us^2 + b) <= f(k) / f(m); its quotients by b and by us are products
 *              with ONE reciprocal each (rb = 1 / b, rus = 1 / us: round 4)
 * Uniforms: Philox4x32-10, counter (call, row_i, row_j, TAG_PERM2X2): trial t = 0, 1, ... (a BTRS candidate or an
 * inversion run) takes words (0, 1) of call t / 2 when t is even and words (2, 3) of the same call when t is odd
 * (round 4: half the calls) — (w + 0.5) 2^-32 each for BTRS, 52 bits + half an ulp for the inversion. */
static uint32_t binom_draw(const perm_tables* t, uint32_t n, uint64_t thr, uint32_t row_i, uint32_t row_j,
                           uint32_t k0, uint32_t k1)
{
    uint32_t out[4], trip = 0, k, tt;
    int flip;
    double p, q, np;
    if (thr == 0 || n == 0) return 0;
    if (thr >= 4294967296ull) return n;
    flip = thr > 2147483648ull;
    tt = flip ? (uint32_t)(4294967296ull - thr) : (uint32_t)thr;
    p = (double)tt * 2.3283064365386963e-10;
    q = 1.0 - p;
    np = (double)n * p;
    philox(0, row_i, row_j, TAG_PERM2X2, k0, k1, out);
    if (np < 10.0) {
        const double qn = lgo_det_exp((double)n * lgo_det_log(q));
        const double lim = np + 10.0 * lgo_det_sqrt(np * q + 1.0);
        const uint32_t bound = lim < (double)n ? (uint32_t)lim : n;
        const double pq = p / q;
        for (;;) {
            const uint32_t w0 = out[(trip & 1u) * 2u], w1 = out[(trip & 1u) * 2u + 1u];
            double U = ((double)(((uint64_t)w0 << 20) | (w1 >> 12)) + 0.5) * 2.220446049250313e-16, T = qn;
            uint32_t x = 0;
            while (U > T && x <= bound) {
                ++x;
                U = (U - T) * (double)x;
                T = T * ((double)(n - x + 1u) * pq);
            }
            if (x <= bound) { k = x; break; }
            if (!(++trip & 1u)) philox(trip >> 1, row_i, row_j, TAG_PERM2X2, k0, k1, out);
        }
    } else {
        const double spq = lgo_det_sqrt(np * q);
        const double b = 1.15 + 2.53 * spq;
        const double a = -0.0873 + 0.0248 * b + 0.01 * p;
        const double c = np + 0.5;
        const double rb = 1.0 / b;
        const double vr = 0.92 - 4.2 * rb;
        const double alpha = (2.83 + 5.1 * rb) * spq;
        const uint32_t m = (uint32_t)floor((double)(n + 1u) * p);
        const double lr = lgo_det_log(p / q);
        const double hm = t->LF[m] + t->LF[n - m];
        for (;;) {
            const uint32_t w0 = out[(trip & 1u) * 2u], w1 = out[(trip & 1u) * 2u + 1u];
            const double u = ((double)w0 + 0.5) * 2.3283064365386963e-10 - 0.5;
            double v = ((double)w1 + 0.5) * 2.3283064365386963e-10;
            const double us = 0.5 - fabs(u);
            const double rus = 1.0 / us;
            const double kf = floor((2.0 * a * rus + b) * u + c);
            if (kf >= 0.0 && kf <= (double)n) {
                double h;
                k = (uint32_t)kf;
                if (us >= 0.07 && v <= vr) break;
                v = v * alpha / (a * rus * rus + b);
                h = hm - t->LF[k] - t->LF[n - k] + ((double)k - (double)m) * lr;
                if (v <= lgo_det_exp(h)) break;
            }
            if (!(++trip & 1u)) philox(trip >> 1, row_i, row_j, TAG_PERM2X2, k0, k1, out);
        }
    }
    return flip ? n - k : k;
}

/* HRUA stays exact with any hat width at least as large as Stadlober's D1 sqrt(var + 1/2) + D2 (the acceptance
 * region only has to lie inside the rectangle the candidates are uniform on).  In a 3 x 2 / 2 x 3 row the draw
 * after the table draw has var = (pop2 - m) m cvar with, over the first draw's window [xlo, xhi], either m or the
 * minority count behind cvar moving monotonically: the largest value over the window serves every shuffle and the
 * square root is taken once per row (times 1 + 2^-40 against rounding). */
static void hrua_width_bound(first_table* ft, uint32_t N, int nr, uint32_t R0, uint32_t R1, uint32_t C0, uint32_t C1)
{
    const uint32_t xlo = ft->klo, xhi = ft->klo + ft->n - 1;
    double varmax;
    if (nr == 3) {
        const uint32_t pop2 = N - R0, good = R1, bad = pop2 - good, mn = good < bad ? good : bad, half = pop2 / 2;
        const uint32_t s_lo = C0 - xhi, s_hi = C0 - xlo;
        const uint32_t m_lo = s_lo < pop2 - s_lo ? s_lo : pop2 - s_lo, m_hi = s_hi < pop2 - s_hi ? s_hi : pop2 - s_hi;
        const uint32_t m_max = (s_lo <= half && half <= s_hi) ? half : (m_lo > m_hi ? m_lo : m_hi);
        const double rp = 1.0 / (double)pop2, rp1 = 1.0 / (double)(pop2 - 1);
        const double d4 = (double)mn * rp, cvar = d4 * (1.0 - d4) * rp1;
        varmax = (double)(pop2 - m_max) * (double)m_max * cvar;
        ft->pop2 = pop2; ft->key2 = good; ft->nr3 = 1;
    } else {
        const uint32_t pop2 = N - C0, m = C1 < pop2 - C1 ? C1 : pop2 - C1, half = pop2 / 2;
        const uint32_t g_lo = R0 - xhi, g_hi = R0 - xlo;
        const uint32_t n_lo = g_lo < pop2 - g_lo ? g_lo : pop2 - g_lo, n_hi = g_hi < pop2 - g_hi ? g_hi : pop2 - g_hi;
        const uint32_t mn_max = (g_lo <= half && half <= g_hi) ? half : (n_lo > n_hi ? n_lo : n_hi);
        const double rp = 1.0 / (double)pop2, rp1 = 1.0 / (double)(pop2 - 1);
        const double d4 = (double)mn_max * rp, cvar = d4 * (1.0 - d4) * rp1;
        varmax = (double)(pop2 - m) * (double)m * cvar;
        ft->pop2 = pop2; ft->key2 = C1; ft->nr3 = 0;
    }
    ft->d7_next = lgo_det_sqrt(varmax + 0.5) * (1.0 + 9.094947017729282e-13);   /* 1 + 2^-40 */
}

/* ------------------------------------------------------------------ 3 x 2 / 2 x 3 rows: 64 lock-step candidate streams
 *
 * Round 3.  A table with six non-empty cells has two real draws: x0 (first row, first column) from the row's
 * threshold table, then one HRUA draw whose parameters depend on x0.  When that second draw is HRUA-sized for every
 * x0 of the table's window (simple_row below — the usual case) the row's n_shuffles tables are produced by 64
 * candidate streams that advance in lock-step "trips" instead of by n_shuffles independent per-shuffle streams:
 *
 *   X[k], k = 0, 1, ...   the first draws: word k & 3 of Philox(counter = (k >> 2, row_i, row_j, TAG_LSX)) through
 *                         the row's threshold table (inverse CDF, as in hg_draw);
 *   stream l = 0 .. 63    candidate pairs (wx, wy): words (0, 1) of Philox((c, row_i, row_j, TAG_LSC + l)) in trip
 *                         2c, words (2, 3) in trip 2c + 1.  u = wx + 1/2, v = (wy - 2^31) + 1/2 (both exact), the
 *                         candidate is w = d6 + d8 v / u (the 2^-32 scalings of Stadlober's (y - 1/2) / x cancel) and
 *                         it is accepted iff 0 <= w < d11 and (u 2^-32)^2 <= exp(t(floor w)), as in hg_draw.
 *   Stream l starts with X[l]; `next` = 64.  In trip t every stream tests one candidate for its current x0.  The
 *   streams that accept complete a table each; taken in stream order they are tables number total, total + 1, ...:
 *   a table with number < n_shuffles counts (exceed += S >= S_obs), the others are surplus of the last trip and are
 *   dropped; the a-th accepting stream of the trip goes on with X[next + a].  Then next and total grow by the number of
 *   acceptances, and the row ends after the trip in which total reaches n_shuffles.
 *
 * Exactly n_shuffles tables are scored; each is an exact draw from the null (an x0 from its marginal, then a
 * rejection-sampled second cell given x0; which tables count depends on acceptance events only, never on the
 * accepted values).  Why: on the GPU the 64 lanes of a wave are the 64 streams.  All lanes need a candidate pair in
 * every trip and none needs a first draw, so one Philox call serves two trips of every lane (round 2: one call per
 * trip, half of its words unused, the table search under a 72 % mask inside the loop), the first draws are made in bulk
 * with all lanes busy and all four words of a call used, nothing is handed out through atomics, and the wave has
 * no drain tail — the lanes stop together. */
#define TAG_LSX 0x70000000u
#define TAG_LSC 0x71000000u

static int simple_row(const first_table* ft, uint32_t N, uint32_t nr, uint32_t nc, uint32_t R0, uint32_t C0, uint32_t C1)
{
    uint32_t xlo, xhi, pop2;
    if (!ft->valid || nr * nc != 6 || !(ft->d7_next > 0.0)) return 0;
    xlo = ft->klo; xhi = ft->klo + ft->n - 1;
    if (nr == 3) {
        pop2 = N - R0;
        return (C0 - xhi >= 10u) && (pop2 - (C0 - xlo) >= 10u) && (C0 - xlo < pop2);
    } else {
        const uint32_t m2 = C1 < (N - C0) - C1 ? C1 : (N - C0) - C1;
        pop2 = N - C0;
        return (m2 >= 10u) && (R0 - xhi >= 1u) && (R0 - xlo < pop2);
    }
}

typedef struct { uint32_t x0, good, sample, m, mn, mx; double d6, d10, d11; } ls_par;

static uint32_t ls_first(const first_table* ft, uint32_t row_i, uint32_t row_j, uint32_t k0, uint32_t k1, uint32_t k)
{
    uint32_t buf[4], lo = 0, hi = ft->n - 1, u;
    philox(k >> 2, row_i, row_j, TAG_LSX, k0, k1, buf);
    u = buf[k & 3u];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (u < ft->thr[mid]) hi = mid; else lo = mid + 1;
    }
    return ft->klo + lo;
}

static void ls_setup(const perm_tables* t, ls_par* p, uint32_t x0, int nr3, uint32_t pop2, uint32_t R0, uint32_t R1,
                     uint32_t C0, uint32_t C1, double d7)
{
    /* the expressions of hg_draw's HRUA branch with the row's hat width d7 */
    const double rp = 1.0 / (double)pop2, rp2 = 1.0 / ((double)pop2 + 2.0);
    uint32_t bad, d9;
    double d4, c9, cap, lim;
    p->x0 = x0;
    if (nr3) { p->good = R1; p->sample = C0 - x0; } else { p->good = R0 - x0; p->sample = C1; }
    bad = pop2 - p->good;
    p->m = p->sample < pop2 - p->sample ? p->sample : pop2 - p->sample;
    p->mn = p->good < bad ? p->good : bad;
    p->mx = p->good < bad ? bad : p->good;
    d4 = (double)p->mn * rp;
    c9 = (double)(p->mn + 1u) * rp2;
    p->d6 = (double)p->m * d4 + 0.5;
    d9 = (uint32_t)floor((double)(p->m + 1u) * c9);
    p->d10 = t->LF[d9] + t->LF[p->mn - d9] + t->LF[p->m - d9] + t->LF[p->mx - p->m + d9];
    cap = (double)((p->m < p->mn ? p->m : p->mn) + 1u);
    lim = floor(p->d6 + 16.0 * d7);
    p->d11 = cap < lim ? cap : lim;
}

static uint32_t perm_lockstep(const perm_tables* t, const first_table* ft, uint32_t N, int nr3, uint32_t R0, uint32_t R1,
                              uint32_t R2, uint32_t C0, uint32_t C1, int64_t sobs, uint32_t row_i, uint32_t row_j,
                              uint32_t n_shuffles, uint32_t k0, uint32_t k1, uint32_t* rec /* (x0, z) of every scored table, or NULL */)
{
    const uint32_t pop2 = nr3 ? N - R0 : N - C0;
    const double d7 = ft->d7_next, d8 = HRUA_D1 * d7 + HRUA_D2;
    ls_par par[64];
    uint32_t buf[64][4];
    uint32_t l, trip, next = 64, total = 0, exceed = 0;
    for (l = 0; l < 64; ++l) ls_setup(t, &par[l], ls_first(ft, row_i, row_j, k0, k1, l), nr3, pop2, R0, R1, C0, C1, d7);
    for (trip = 0; total < n_shuffles; ++trip) {
        uint32_t acc = 0;
        for (l = 0; l < 64; ++l) {
            ls_par* p = &par[l];
            uint32_t wx, wy, zc, z;
            double u, v, w, tt, x;
            if ((trip & 1u) == 0) philox(trip >> 1, row_i, row_j, TAG_LSC + l, k0, k1, buf[l]);
            wx = buf[l][2 * (trip & 1u)];
            wy = buf[l][2 * (trip & 1u) + 1];
            u = (double)wx + 0.5;
            v = (double)(int32_t)(wy ^ 0x80000000u) + 0.5;
            w = p->d6 + d8 * v / u;
            if (w < 0.0 || w >= p->d11) continue;
            zc = (uint32_t)floor(w);
            tt = p->d10 - (t->LF[zc] + t->LF[p->mn - zc] + t->LF[p->m - zc] + t->LF[p->mx - p->m + zc]);
            x = u * 2.3283064365386963e-10;                 /* 2^-32 */
            if (!(x * x <= lgo_det_exp(tt))) continue;
            z = zc;
            if (p->good > pop2 - p->good) z = p->m - z;     /* z counted the minority kind */
            if (p->m < p->sample) z = p->good - z;          /* drew the complement */
            if (total + acc < n_shuffles) {
                const uint32_t x0 = p->x0;
                int64_t ss;
                if (rec) { rec[2 * (total + acc)] = x0; rec[2 * (total + acc) + 1] = z; }
                if (nr3) {
                    const uint32_t x2 = C0 - x0 - z;
                    ss = t->G[x0] + t->G[z] + t->G[x2] + t->G[R0 - x0] + t->G[R1 - z] + t->G[R2 - x2];
                } else {
                    ss = t->G[x0] + t->G[C0 - x0] + t->G[z] + t->G[C1 - z] + t->G[R0 - x0 - z] +
                         t->G[R1 - (C0 - x0) - (C1 - z)];
                }
                exceed += (ss >= sobs);
            }
            ls_setup(t, p, ls_first(ft, row_i, row_j, k0, k1, next + acc), nr3, pop2, R0, R1, C0, C1, d7);
            acc++;
        }
        next += acc;
        total += acc;
    }
    return exceed;
}

/* ------------------------------------------------------------------ small tables: exact mass + one binomial variate
 * (round 3)  The number of shuffles with S >= S_obs is Binomial(n_shuffles, P) whatever the table's shape; for 2 x 2
 * tables P comes from ptail22.  A larger table whose margins admit few tables gets its P by ENUMERATION instead of
 * n_shuffles Monte-Carlo tables — rows of a few hundred reads with a rare third allele, the shape real footprints have,
 * are a few hundred tables:
 *   la, lb   = the (first) largest row margin, the (first) largest column margin: their cells are the dependent ones
 *   free     = the four cells (a, b), a != la, b != lb, in row-major order; cell (a, b) runs over 0 .. min(R[a], C[b])
 *   n_tables = the product of the four (min(R[a], C[b]) + 1); the row is enumerated iff
 *              n_tables <= ENUM_MAX and n_tables <= 4 n_shuffles
 *   table t  = mixed-radix digits of t (first free cell = least significant); the dependent cells follow from the
 *              margins; the table exists iff none of them is negative
 *   mass     = sum over the existing tables with S >= S_obs of trunc(2^62 exp(c0 - lf)); lf = the nine LF[cell] added
 *              row by row with the rows taken in the order (free, free, la) and the columns in the order (free, free, lb) —
 *              the layout in which the GPU holds the table; c0 = LF[R0] + LF[R1] + LF[R2] + LF[C0] + LF[C1] + LF[C2] - LF[N]
 *              added in this order.  The mass is an integer sum: any order
 *   exceed   = binom_draw(n_shuffles, min(2^32, mass >> 30), ...)      (the 2 x 2 rows' stream: a row is one or the other)
 */
static uint32_t ENUM_MAX = 4096u;   /* (a variable only so that tests can switch the enumeration off: lgo_set_enum_max) */
static int enum_plan_x(const uint32_t R[3], const uint32_t C[3], uint32_t n_shuffles, int exact, uint32_t fa[2], uint32_t fb[2],
                       uint32_t* la_out, uint32_t* lb_out, uint32_t radix[4], uint64_t* n_tables)
{
    uint32_t la = 0, lb = 0, a, k = 0;
    uint64_t n = 1;
    for (a = 1; a < 3; ++a) { if (R[a] > R[la]) la = a; if (C[a] > C[lb]) lb = a; }
    for (a = 0, k = 0; a < 3; ++a) if (a != la) fa[k++] = a;
    for (a = 0, k = 0; a < 3; ++a) if (a != lb) fb[k++] = a;
    for (k = 0; k < 4; ++k) {
        const uint32_t ra = R[fa[k >> 1]], cb = C[fb[k & 1]];
        radix[k] = (ra < cb ? ra : cb) + 1u;
        n *= radix[k];
        if (n > ENUM_MAX) return 0;
    }
    *la_out = la; *lb_out = lb; *n_tables = n;
    return exact || n <= 4ull * (uint64_t)n_shuffles;      /* (the exact-p entry enumerates whatever fits ENUM_MAX) */
}
static int enum_plan(const uint32_t R[3], const uint32_t C[3], uint32_t n_shuffles, uint32_t fa[2], uint32_t fb[2],
                     uint32_t* la_out, uint32_t* lb_out, uint32_t radix[4], uint64_t* n_tables)
{
    return enum_plan_x(R, C, n_shuffles, 0, fa, fb, la_out, lb_out, radix, n_tables);
}

static uint64_t enum_mass(const perm_tables* t, const uint32_t R[3], const uint32_t C[3], uint32_t N, int64_t sobs,
                          const uint32_t fa[2], const uint32_t fb[2], uint32_t la, uint32_t lb, const uint32_t radix[4],
                          uint64_t n_tables)
{
    uint64_t mass = 0, tt;
    double c0 = t->LF[R[0]];
    c0 += t->LF[R[1]]; c0 += t->LF[R[2]]; c0 += t->LF[C[0]]; c0 += t->LF[C[1]]; c0 += t->LF[C[2]]; c0 -= t->LF[N];
    for (tt = 0; tt < n_tables; ++tt) {
        int64_t x[3][3], rest;
        uint32_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, k;
        uint64_t d = tt;
        int ok = 1;
        double lf;
        for (k = 0; k < 4; ++k) { x[fa[k >> 1]][fb[k & 1]] = (int64_t)(d % radix[k]); d /= radix[k]; }
        for (k = 0; k < 2; ++k) {
            x[fa[k]][lb] = (int64_t)R[fa[k]] - x[fa[k]][fb[0]] - x[fa[k]][fb[1]];
            x[la][fb[k]] = (int64_t)C[fb[k]] - x[fa[0]][fb[k]] - x[fa[1]][fb[k]];
        }
        rest = (int64_t)R[la] - x[la][fb[0]] - x[la][fb[1]];
        x[la][lb] = rest;
        for (k = 0; k < 9; ++k) { if (x[k / 3][k % 3] < 0) ok = 0; else T[k] = (uint32_t)x[k / 3][k % 3]; }
        if (!ok || stat9(t, T) < sobs) continue;
        {
            const uint32_t ro[3] = {fa[0], fa[1], la}, co[3] = {fb[0], fb[1], lb};
            lf = t->LF[T[3 * ro[0] + co[0]]];
            for (k = 1; k < 9; ++k) lf += t->LF[T[3 * ro[k / 3] + co[k % 3]]];
        }
        mass += (uint64_t)(lgo_det_exp(c0 - lf) * 4611686018427387904.0);
    }
    return mass;
}

/* ------------------------------------------------------------------ six-cell tables: exact mass chord by chord (round 4)
 * A 3 x 2 / 2 x 3 table — a tri-allelic site against a bi-allelic one, 96 % of the larger-than-2x2 rows of a dense
 * chromosome — has two degrees of freedom, and the set {S < S_obs} is the lattice inside a convex curve around the
 * table of independence: for null pairs a few ten thousand tables at 2e5 reads, far fewer than the arithmetic of
 * n_shuffles rejection-sampled tables.  Its mass is summed exactly, chord by chord, each chord being a 2 x 2 problem
 * (unit_mass above), and the exceed count is ONE binomial variate, as for 2 x 2 rows and enumerated rows.
 *
 *   canonical form: three CLASSES with margins A[0..2] (the rows when three rows are non-empty, else the columns, in
 *       table order) against two SIDES with margins B0, B1; a table is (a_0, a_1, a_2) = the classes' counts on side 0,
 *       a_0 + a_1 + a_2 = B0.  o = the class with the (first) smallest margin, p < q the other two in order.
 *       z = a_o is the chord index, x = a_p runs along the chord, a_q = B0 - z - x.
 *   chord z: the 2 x 2 problem h_z = {N' = A_p + A_q, K = A_p, n = B0 - z} in x with
 *       S(x, z)   = stat22(h_z, x) + G[z] + G[A_o - z],      so  S < S_obs  <=>  stat22(h_z, x) < sobs_z = sobs - G[z] - G[A_o - z]
 *       pmf(x, z) = exp(c0_z - LF[x] - LF[K - x] - LF[n - x] - LF[N' - K - n + x]),   c0_z = cJ - LF[z] - LF[A_o - z]  (in this order),
 *       cJ = LF[A_0] + LF[A_1] + LF[A_2] + LF[B0] + LF[B1] - LF[N]  (in this order)
 *       stat22 falls up to kc = floor(n K / N') and rises after it: the chord's inside is (klo_z, khi_z) with
 *       klo_z = the last x in [kmin - 1, kc] with stat22 >= sobs_z (kmin - 1: none), khi_z = the first x in [kc + 1, kmax + 1]
 *       with stat22 >= sobs_z (kmax + 1: none); both boundaries of monotone predicates (any search order finds them).
 *   which chords: min over real x of S(x, z) = stat22(h_c, z) + G[A_p] + G[A_q] - G[A_p + A_q] with the COLLAPSED table
 *       h_c = {N, K = A_o, n = B0} (class o against the rest) — convex in z, and below the chord's lattice minimum (G is
 *       rounded: 8 units of slack).  Chords z in (zlo, zhi), the boundaries of stat22(h_c, z) >= sc = sobs + 8 - (G[A_p] + G[A_q]
 *       - G[A_p + A_q]) on either side of floor(B0 A_o / N), found like klo / khi; a chord there may still be empty.
 *   zero test first: ln n! >= n ln n - n, so every table with S >= S_obs has pmf <= exp(cJ + N - sobs 2^-28) and there
 *       are at most (A_o + 1)(A_p + 1) of them: if cJ + N + ln((A_o + 1)(A_p + 1)) - sobs 2^-28 < -23.1 the set weighs less
 *       than 2^-33, thr = 0 (linked pairs: a tri-allelic het SNP against another het SNP).
 *   gate: the work is proportional to the region's perimeter (six_inside_walk below): box = (zhi - zlo - 1) + max(1, length
 *       of the chord through z_c = floor(B0 A_o / N) clamped into (zlo, zhi)); the row takes this path iff
 *       box <= min(2^22, SIX_PTS * n_shuffles / 16) — with SIX_PTS = 16: as many chords as shuffles, about where the walk
 *       costs what n_shuffles sampled tables cost; otherwise (and all 3 x 3 rows) the Monte-Carlo paths below.
 *   inside = the mass of {S < S_obs}: six_inside_walk (perimeter); six_inside (the plain sum over the area, chord by chord
 *       in units of 64 values) is kept as the independent check the tests hold it against,
 *   thr = (2^62 - inside) >> 30 (0 when inside > 2^62), exceed = binom_draw(n_shuffles, thr, ...).
 * Accuracy: as the 2 x 2 centre form — |dP| <= 1e-9 up to 2e5 reads (the LF table's rounding times the inside mass).
 */
static uint32_t SIX_PTS = 16u;     /* sixteenths of a chord per shuffle a row may cost (0: the path is off; lgo_set_six_pts) */

typedef struct {
    uint32_t A[3], B0, B1, N, o, p, q;     /* canonical margins, o / p / q = indices into A */
    int64_t sobs;
    double cJ;
    hg22 hc;                                /* the collapsed table */
    int64_t zlo, zhi;                       /* chords z in (zlo, zhi) */
    int zero;                               /* the zero test fired: thr = 0 */
    double zero_bound;                      /* ... and the bound it fired on: ln of an upper bound of the tail mass */
} six_row;

/* last x in [lo, hi] with stat22(h, x) >= s for a predicate that is true then false along x; lo is a sentinel (never evaluated) */
static int64_t last_ge(const perm_tables* t, const hg22* h, int64_t lo, int64_t hi, int64_t s)
{
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo + 1) / 2;
        if (stat22(t, h, (uint32_t)mid) >= s) lo = mid; else hi = mid - 1;
    }
    return lo;
}
/* first x in [lo, hi] with stat22(h, x) >= s, false then true; hi is a sentinel */
static int64_t first_ge(const perm_tables* t, const hg22* h, int64_t lo, int64_t hi, int64_t s)
{
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (stat22(t, h, (uint32_t)mid) >= s) hi = mid; else lo = mid + 1;
    }
    return lo;
}

static void hg22_set(hg22* h, uint32_t N, uint32_t K, uint32_t n, double c0)
{
    h->N = N; h->K = K; h->n = n;
    h->kmin = K + n > N ? K + n - N : 0;
    h->kmax = K < n ? K : n;
    h->c0 = c0;
}

static uint32_t kc_of(const hg22* h)
{
    uint32_t kc = (uint32_t)(((uint64_t)h->n * (uint64_t)h->K) / (uint64_t)h->N);
    if (kc < h->kmin) kc = h->kmin;
    if (kc > h->kmax) kc = h->kmax;
    return kc;
}

/* chord z of a six-cell row: the 2 x 2 problem and the inside (klo, khi) */
static void six_chord(const perm_tables* t, const six_row* s, int64_t z, hg22* h, int64_t* klo, int64_t* khi)
{
    const uint32_t Ao = s->A[s->o], Ap = s->A[s->p], Aq = s->A[s->q];
    const int64_t sobs_z = s->sobs - t->G[z] - t->G[Ao - z];
    double c0 = s->cJ;
    uint32_t kc;
    c0 -= t->LF[z];
    c0 -= t->LF[Ao - z];
    hg22_set(h, Ap + Aq, Ap, (uint32_t)(s->B0 - z), c0);
    kc = kc_of(h);
    *klo = last_ge(t, h, (int64_t)h->kmin - 1, (int64_t)kc, sobs_z);
    *khi = first_ge(t, h, (int64_t)kc + 1, (int64_t)h->kmax + 1, sobs_z);
}

/* returns 1 when the row takes the six-cell path at this n_shuffles */
static int six_plan_x(const perm_tables* t, const uint32_t T[9], uint32_t n_shuffles, int exact, six_row* s);
static int six_plan(const perm_tables* t, const uint32_t T[9], uint32_t n_shuffles, six_row* s)
{
    return six_plan_x(t, T, n_shuffles, 0, s);
}
/* exact: the exact-p entry (lgmi_params.exact_2x2): the gate is 2^20 chords + central chord length whatever n_shuffles is */
static int six_plan_x(const perm_tables* t, const uint32_t T[9], uint32_t n_shuffles, int exact, six_row* s)
{
    uint32_t R[3], C[3], nzr[3], nzc[3], nr = 0, nc = 0, a, N = 0;
    uint64_t box_max, box;
    int64_t sc, zc, klo, khi, len;
    hg22 h;
    if (!SIX_PTS) return 0;
    for (a = 0; a < 3; ++a) { R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2]; C[a] = T[a] + T[3 + a] + T[6 + a]; N += R[a]; }
    for (a = 0; a < 3; ++a) { if (R[a]) nzr[nr++] = a; if (C[a]) nzc[nc++] = a; }
    if (nr * nc != 6) return 0;
    if (nr == 3) { for (a = 0; a < 3; ++a) s->A[a] = R[a]; s->B0 = C[nzc[0]]; s->B1 = C[nzc[1]]; }
    else { for (a = 0; a < 3; ++a) s->A[a] = C[a]; s->B0 = R[nzr[0]]; s->B1 = R[nzr[1]]; }
    s->N = N;
    s->o = 0;
    if (s->A[1] < s->A[s->o]) s->o = 1;
    if (s->A[2] < s->A[s->o]) s->o = 2;
    s->p = s->o == 0 ? 1 : 0;
    s->q = s->o == 2 ? 1 : 2;
    s->sobs = stat9(t, T);
    s->cJ = t->LF[s->A[0]];
    s->cJ += t->LF[s->A[1]]; s->cJ += t->LF[s->A[2]]; s->cJ += t->LF[s->B0]; s->cJ += t->LF[s->B1]; s->cJ -= t->LF[N];
    s->zero = 0; s->zlo = 0; s->zhi = 0;
    {
        const double lnt = lgo_det_log(((double)s->A[s->o] + 1.0) * ((double)s->A[s->p] + 1.0));
        double b = s->cJ;
        b += (double)N;
        b += lnt;
        b -= (double)s->sobs * 3.725290298461914e-09;     /* 2^-28 */
        if (b < -23.1) { s->zero = 1; s->zero_bound = b; return 1; }
    }
    hg22_set(&s->hc, N, s->A[s->o], s->B0, 0.0);
    sc = s->sobs + 8 - (t->G[s->A[s->p]] + t->G[s->A[s->q]] - t->G[s->A[s->p] + s->A[s->q]]);
    zc = (int64_t)kc_of(&s->hc);
    s->zlo = last_ge(t, &s->hc, (int64_t)s->hc.kmin - 1, zc, sc);
    s->zhi = first_ge(t, &s->hc, zc + 1, (int64_t)s->hc.kmax + 1, sc);
    if (s->zhi - s->zlo - 1 <= 0) return 1;                 /* no table below S_obs: inside = 0, thr = 2^32 */
    if (zc <= s->zlo) zc = s->zlo + 1;
    if (zc >= s->zhi) zc = s->zhi - 1;
    six_chord(t, s, zc, &h, &klo, &khi);
    len = khi - klo - 1;
    box = (uint64_t)(s->zhi - s->zlo - 1) + (uint64_t)(len > 1 ? len : 1);
    box_max = ((uint64_t)SIX_PTS * (uint64_t)n_shuffles) >> 4;
    if (box_max > 4194304ull) box_max = 4194304ull;         /* 2^22 */
    if (exact) box_max = 1048576ull;                        /* 2^20 */
    return box <= box_max;
}

/* the mass of {S < S_obs} in units of 2^-62 (integer sum over the chords' units), the number of lattice points */
static uint64_t six_inside(const perm_tables* t, const six_row* s, uint64_t* area)
{
    uint64_t m = 0, ar = 0;
    int64_t z;
    for (z = s->zlo + 1; z < s->zhi; ++z) {
        hg22 h;
        int64_t klo, khi;
        six_chord(t, s, z, &h, &klo, &khi);
        if (khi - klo - 1 > 0) { m += range_mass(t, &h, klo + 1, khi - klo - 1); ar += (uint64_t)(khi - klo - 1); }
    }
    if (area) *area = ar;
    return m;
}

/* The joint pmf J(x) = pmf22(h, x) along lo .. hi by unit_mass's recurrence (first term from the log-factorials, sub-blocks
 * of SUB steps; any length), J = 0 outside the chord's support [kmin, kmax]:
 *     *first = J(lo), *last = J(hi), *rest = the sum of J over lo + 1 .. hi.
 * With lo' = max(lo, kmin), hi' = min(hi, kmax) (nothing when lo' > hi'): the recurrence starts at lo' with term0 = pmf22(lo')
 * and sums the terms after it into rest'; first = term0 and rest = rest' when lo >= kmin, else first = 0 and
 * rest = term0 + rest'; last = the recurrence's final term when hi <= kmax, else 0. */
static void walk_sum(const perm_tables* t, const hg22* h, int64_t lo, int64_t hi, double* first, double* last, double* rest)
{
    const int64_t lo2 = lo > (int64_t)h->kmin ? lo : (int64_t)h->kmin, hi2 = hi < (int64_t)h->kmax ? hi : (int64_t)h->kmax;
    uint32_t k;
    double term0, term, sum = 0.0;
    int64_t rem;
    *first = 0.0; *last = 0.0; *rest = 0.0;
    if (lo2 > hi2) return;
    k = (uint32_t)lo2;
    term0 = pmf22(t, h, k);
    term = term0;
    rem = hi2 - lo2;
    while (rem > 0) {
        const int64_t m = rem < SUB ? rem : SUB;
        double P = 0.0, Nn = 1.0, Q = 1.0;
        double a = (double)(h->K - k), b = (double)(h->n - k), c = (double)(k + 1u), d = (double)(h->N - h->K - h->n + k + 1u);
        int64_t j;
        for (j = 0; j < m; ++j) {
            const double num = a * b, den = c * d;
            Nn = Nn * num;
            Q = Q * den;
            P = fma(P, den, Nn);
            a -= 1.0; b -= 1.0; c += 1.0; d += 1.0;
        }
        {
            const double rQ = 1.0 / Q;
            sum += (term * P) * rQ;
            term = (term * Nn) * rQ;
        }
        k += (uint32_t)m;
        rem -= m;
    }
    if (lo >= (int64_t)h->kmin) { *first = term0; *rest = sum; } else { *rest = term0 + sum; }
    if (hi <= (int64_t)h->kmax) *last = term;
}

/* The mass of {S < S_obs} along the PERIMETER instead of over the area.  With X_n ~ HG(N', K, n) one more draw gives
 * P(X_{n+1} <= x) = P(X_n <= x) - P(X_n = x) (K - x) / (N' - n); chord z + 1 has n - 1 draws, so for a fixed range (a, b] of x
 *     M'(z + 1) = rho_z M(z) + [J(b, z + 1) (K - b) - J(a, z + 1) (K - a)] / (N' - n_{z+1}),
 *     rho_z = w(z + 1) / w(z) = (A_o - z)(n_z) / ((z + 1)(N' - n_z + 1))          (w = the collapsed table's pmf)
 * with M(z) = the joint mass of chord z over (a, b] and J the joint pmf (0 outside the chord's support): exact, a few terms per
 * chord instead of the chord's length.  Then the range is moved from the previous chord's inside (a_p, b_p] to this chord's
 * (a, b] by adding / removing the few values in between.  So every chord is an affine map of its predecessor's mass,
 * M_c = rho_c M_p + beta_c, whose coefficients depend on the two chords' bounds only.  Chord j = 0, 1, .. of the row, z = zlo + 1 + j,
 * n = B0 - z, K = A_p, N' = A_p + A_q, (a, b] = (klo, khi - 1] its inside, (a_p, b_p] its predecessor's (j = 0: a_p = b_p = its own kc):
 *     top      walk_sum(min(b, b_p) .. max(b, b_p)) -> first, last, rest:   Jb = b >= b_p ? first : last,   adjT = b >= b_p ? rest : -rest
 *     bottom   walk_sum(min(a, a_p) .. max(a, a_p)) -> first, last, rest:   Ja = a <  a_p ? last : first,   adjB = a <  a_p ? rest : -rest
 *     rzd    = 1 / ((double)z (double)(N' - n));   rho = ((double)(A_o - z + 1) (double)(n + 1)) rzd
 *     t2     = (Jb (double)(K - b_p) - Ja (double)(K - a_p)) ((double)z rzd);   beta = (t2 + adjT) + adjB
 *     j = 0: rho = 0, t2 = 0 (the adjustments sum the whole chord);   an empty chord (b <= a): rho = beta = 0.
 * The maps are composed 16 chords at a time (sub-chunk g = chords 16 g .. 16 g + 15 of the row): an inclusive Hillis-Steele scan
 * inside the sub-chunk (offsets 1, 2, 4, 8: beta <- rho beta_prev + beta, rho <- rho rho_prev, products and sums rounded
 * separately), applied to the mass of the previous sub-chunk's last chord (0 for g = 0): M_j = rho_incl M_carry + beta_incl.
 * A chord's composite depends on its distance from the sub-chunk's start only, so the GPU may lay sub-chunks of different
 * rows side by side in one wave.  inside = sum over the chords of trunc(2^62 min(max(M_j, 0), 1)): an integer sum, any order.
 * All of it is IEEE +, -, *, / on doubles in a fixed order: the same bits on both sides. */
static uint64_t six_inside_walk(const perm_tables* t, const six_row* s)
{
    const uint32_t Ao = s->A[s->o], Ap = s->A[s->p], Aq = s->A[s->q], Np = Ap + Aq, K = Ap;
    const int64_t nz = s->zhi - s->zlo - 1;
    uint64_t inside = 0;
    double M_carry = 0.0;
    int64_t a_carry = 0, b_carry = 0, base;
    for (base = 0; base < nz; base += 16) {
        double rho[16], beta[16];
        int64_t av[16], bv[16];
        const int cnt = (int)(nz - base < 16 ? nz - base : 16);
        int l, o;
        for (l = 0; l < 16; ++l) { rho[l] = 1.0; beta[l] = 0.0; av[l] = bv[l] = 0; }
        for (l = 0; l < cnt; ++l) {
            const int64_t z = s->zlo + 1 + base + l;
            const int first = (base == 0 && l == 0);
            hg22 h;
            int64_t klo, khi, a, b, a_p, b_p;
            double r = 0.0, t2 = 0.0, adjT, adjB, be, fT, lT, rT, fB, lB, rB;
            six_chord(t, s, z, &h, &klo, &khi);
            a = klo; b = khi - 1;
            av[l] = a; bv[l] = b;
            if (first) { a_p = b_p = (int64_t)kc_of(&h); }
            else if (l == 0) { a_p = a_carry; b_p = b_carry; }
            else { a_p = av[l - 1]; b_p = bv[l - 1]; }
            walk_sum(t, &h, b < b_p ? b : b_p, b < b_p ? b_p : b, &fT, &lT, &rT);
            walk_sum(t, &h, a < a_p ? a : a_p, a < a_p ? a_p : a, &fB, &lB, &rB);
            adjT = b >= b_p ? rT : -rT;
            adjB = a < a_p ? rB : -rB;
            if (!first) {
                const double Jb = b >= b_p ? fT : lT, Ja = a < a_p ? lB : fB;
                const double zd = (double)z;
                const double rzd = 1.0 / (zd * (double)(Np - h.n));
                const double f = (double)(Ao - z + 1) * (double)(h.n + 1u);
                double tb, ta;
                r = f * rzd;
                tb = Jb * (double)((int64_t)K - b_p);
                ta = Ja * (double)((int64_t)K - a_p);
                t2 = tb - ta;
                t2 = t2 * (zd * rzd);
            }
            be = t2 + adjT;
            be = be + adjB;
            if (b - a <= 0) { r = 0.0; be = 0.0; }
            rho[l] = r; beta[l] = be;
        }
        for (o = 1; o < 16; o <<= 1) {                   /* inclusive scan of the affine maps */
            double pr[16], pb[16];
            for (l = 0; l < 16; ++l) { pr[l] = rho[l]; pb[l] = beta[l]; }
            for (l = o; l < 16; ++l) {
                const double x = pr[l] * pb[l - o];
                beta[l] = x + pb[l];
                rho[l] = pr[l] * pr[l - o];
            }
        }
        for (l = 0; l < cnt; ++l) {
            double x = rho[l] * M_carry;
            x = x + beta[l];
            rho[l] = x;                                   /* (rho[] now holds the chords' masses) */
            if (x > 0.0) inside += x >= 1.0 ? 4611686018427387904ull : (uint64_t)(x * 4611686018427387904.0);
        }
        M_carry = rho[cnt - 1]; a_carry = av[cnt - 1]; b_carry = bv[cnt - 1];
    }
    return inside;
}

static uint64_t six_thr(const perm_tables* t, const six_row* s, double* p_out)
{
    uint64_t inside;
    /* the exact p of a row the zero test decided: the BOUND the test fired on (an upper bound of the tail mass, below
     * e^-23.1 = 9.3e-11), floored at the smallest normal double — never 0.0, which no permutation test can give (the observed
     * table is itself in the tail) and which breaks -log10(p) downstream (advice r4) */
    if (s->zero) {
        if (p_out) { const double pb = lgo_det_exp(s->zero_bound); *p_out = pb > 2.2250738585072014e-308 ? pb : 2.2250738585072014e-308; }
        return 0;
    }
    inside = six_inside_walk(t, s);
    if (p_out) { *p_out = 1.0 - (double)inside * 2.168404344971009e-19; if (*p_out < 0.0) *p_out = 0.0; }
    return inside <= 4611686018427387904ull ? (4611686018427387904ull - inside) >> 30 : 0;
}

static uint32_t perm_one(const perm_tables* t, const uint32_t T[9], uint32_t row_i, uint32_t row_j,
                         uint32_t n_shuffles, uint64_t seed, double* ptail_out)
{
    uint32_t R[3], C[3], N = 0, nzr[3], nzc[3], nr = 0, nc = 0, a, b, s, exceed = 0;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if (ptail_out) *ptail_out = NAN;
    for (a = 0; a < 3; ++a) {
        R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2];
        C[a] = T[a] + T[3 + a] + T[6 + a];
        N += R[a];
    }
    for (a = 0; a < 3; ++a) { if (R[a]) nzr[nr++] = a; if (C[a]) nzc[nc++] = a; }
    if (nr <= 1 || nc <= 1) { if (ptail_out) *ptail_out = 1.0; return n_shuffles; }
    if (nr == 2 && nc == 2) {
        hg22 h;
        uint64_t thr;
        const uint32_t kobs = T[3 * nzr[1] + nzc[1]];
        h.N = N; h.K = R[nzr[1]]; h.n = C[nzc[1]];
        h.kmin = h.K + h.n > N ? h.K + h.n - N : 0;
        h.kmax = h.K < h.n ? h.K : h.n;
        h.c0 = t->LF[h.K];
        h.c0 += t->LF[N - h.K];
        h.c0 += t->LF[h.n];
        h.c0 += t->LF[N - h.n];
        h.c0 -= t->LF[N];
        thr = ptail22(t, &h, kobs, ptail_out);
        return binom_draw(t, n_shuffles, thr, row_i, row_j, k0, k1);
    }
    {
        const int64_t sobs = stat9(t, T);
        first_table ft;
        {
            uint32_t fa[2], fb[2], la, lb, radix[4];
            uint64_t n_tables;
            if (enum_plan_x(R, C, n_shuffles, ptail_out != NULL, fa, fb, &la, &lb, radix, &n_tables)) {
                const uint64_t mass = enum_mass(t, R, C, N, sobs, fa, fb, la, lb, radix, n_tables);
                uint64_t thr = mass >> 30;
                if (thr > 4294967296ull) thr = 4294967296ull;
                if (ptail_out) { *ptail_out = (double)mass * 2.168404344971009e-19; if (*ptail_out > 1.0) *ptail_out = 1.0; return 0; }
                return binom_draw(t, n_shuffles, thr, row_i, row_j, k0, k1);
            }
        }
        {
            /* (the exact-p entry, round 4 late: larger tables whose exact mass is within reach — enumeration above, the
             *  perimeter walk here — return it; the others keep NaN) */
            six_row sx;
            if (six_plan_x(t, T, n_shuffles, ptail_out != NULL, &sx)) {
                if (ptail_out) { (void)six_thr(t, &sx, ptail_out); return 0; }
                return binom_draw(t, n_shuffles, six_thr(t, &sx, NULL), row_i, row_j, k0, k1);
            }
        }
        if (ptail_out) return 0;     /* no exact form within reach: NaN */
        first_table_build(t, N, R[nzr[0]], C[nzc[0]], &ft);   /* the first non-empty row and column */
        if (ft.valid && nr * nc == 6 && N - R[nzr[0]] > 1 && N - C[nzc[0]] > 1)
            hrua_width_bound(&ft, N, (int)nr, R[nzr[0]], R[nzr[1]], C[nzc[0]], C[nzc[1]]);
        if (simple_row(&ft, N, nr, nc, R[nzr[0]], C[nzc[0]], C[nzc[1]]))
            return perm_lockstep(t, &ft, N, nr == 3, R[nzr[0]], R[nzr[1]], nr == 3 ? R[nzr[2]] : 0, C[nzc[0]], C[nzc[1]],
                                 sobs, row_i, row_j, n_shuffles, k0, k1, NULL);
        for (s = 0; s < n_shuffles; ++s) {
            gen_stream g;
            uint32_t rr[3] = {R[0], R[1], R[2]}, Ts[9], pop_all = N;
            g.c0 = s; g.c1 = row_i; g.c2 = row_j; g.k0 = k0; g.k1 = k1; g.call = 0; g.have = 0; g.after_table = 0;
            for (b = 0; b < 3; ++b) {
                /* column b: distribute C[b] reads over the rows' remaining capacities */
                uint32_t cc = C[b], pop = pop_all;
                for (a = 0; a < 3; ++a) {
                    const uint32_t x = hg_draw(t, pop, rr[a], cc, &g, &ft);
                    Ts[3 * a + b] = x;
                    pop -= rr[a];
                    cc -= x;
                }
                for (a = 0; a < 3; ++a) rr[a] -= Ts[3 * a + b];
                pop_all -= C[b];
            }
            exceed += (stat9(t, Ts) >= sobs);
        }
        return exceed;
    }
}

int lgo_perm_rows(uint64_t n_rows, const uint32_t* row_i, const uint32_t* row_j, const uint32_t* counts,
                  uint32_t n_shuffles, uint64_t seed, double* p_out, uint32_t* exceed_out, int n_threads)
{
    perm_tables t;
    uint32_t max_n = 0;
    uint64_t r;
    for (r = 0; r < n_rows; ++r) {
        uint32_t n = 0;
        int k;
        for (k = 0; k < 9; ++k) n += counts[9 * r + k];
        if (n > max_n) max_n = n;
    }
    if (n_shuffles > max_n) max_n = n_shuffles;        /* binom_draw looks up LF[0 .. n_shuffles] */
    if (tables_init(&t, max_n)) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t rr = 0; rr < (int64_t)n_rows; ++rr) {
        const uint32_t e = perm_one(&t, counts + 9 * rr, row_i[rr], row_j[rr], n_shuffles, seed, NULL);
        exceed_out[rr] = e;
        p_out[rr] = (1.0 + (double)e) / ((double)n_shuffles + 1.0);
    }
    free(t.G); free(t.LF);
    return 0;
}

/* exact permutation p of the rows that have one within reach (lgmi_params.exact_2x2): tables with at most 2 x 2 non-empty
 * classes (the mass of the tables with S >= S_obs, summed as in ptail22; 1.0 for degenerate tables); larger tables with at
 * most ENUM_MAX candidate tables (enum_mass); 3 x 2 / 2 x 3 tables whose chords + central chord length are at most 2^20 (one
 * minus six_inside_walk; 0 when the zero test fires).  NaN for the others (they keep the Monte-Carlo estimate). */
int lgo_perm_rows_exact(uint64_t n_rows, const uint32_t* counts, double* p_out)
{
    perm_tables t;
    uint32_t max_n = 0;
    uint64_t r;
    for (r = 0; r < n_rows; ++r) {
        uint32_t n = 0;
        int k;
        for (k = 0; k < 9; ++k) n += counts[9 * r + k];
        if (n > max_n) max_n = n;
    }
    if (tables_init(&t, max_n)) return -1;
    for (r = 0; r < n_rows; ++r) (void)perm_one(&t, counts + 9 * r, 0, 1, 0, 0, &p_out[r]);
    free(t.G); free(t.LF);
    return 0;
}

/* ---- hooks for the statistical tests of this specification (tests/test_perm_oracle.py) ---- */
int lgo_hg_draw_many2(uint32_t pop, uint32_t good, uint32_t sample, uint64_t seed, uint32_t n, uint32_t* out, int use_table)
{
    perm_tables t;
    first_table ft;
    uint32_t i;
    if (tables_init(&t, pop)) return -1;
    ft.valid = 0;
    if (use_table) first_table_build(&t, pop, good, sample, &ft);
    if (use_table && !ft.valid) { free(t.G); free(t.LF); return -2; }
    for (i = 0; i < n; ++i) {
        gen_stream g;
        g.c0 = i; g.c1 = 1; g.c2 = 2; g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.call = 0; g.have = 0; g.after_table = 0;
        out[i] = hg_draw(&t, pop, good, sample, &g, use_table ? &ft : NULL);
    }
    free(t.G); free(t.LF);
    return 0;
}

/* HRUA draws with the hat width inflated by `factor` (the row-constant bound of hrua_width_bound is such an inflation) */
int lgo_hg_draw_wide(uint32_t pop, uint32_t good, uint32_t sample, uint64_t seed, uint32_t n, uint32_t* out, double factor)
{
    perm_tables t;
    first_table* ft = (first_table*)calloc(1, sizeof(first_table));
    uint32_t i;
    const uint32_t bad = pop - good, mn = good < bad ? good : bad, m = sample < pop - sample ? sample : pop - sample;
    if (!ft || tables_init(&t, pop)) { free(ft); return -1; }
    {
        const double rp = 1.0 / (double)pop, rp1 = 1.0 / (double)(pop - 1);
        const double d4 = (double)mn * rp, cvar = d4 * (1.0 - d4) * rp1;
        ft->valid = 1; ft->pop2 = pop; ft->key2 = good; ft->nr3 = 1;
        ft->d7_next = lgo_det_sqrt((double)(pop - m) * (double)m * cvar + 0.5) * factor;
    }
    for (i = 0; i < n; ++i) {
        gen_stream g;
        g.c0 = i; g.c1 = 5; g.c2 = 6; g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32); g.call = 1;
        philox(i, 5, 6, TAG_PERMGEN, g.k0, g.k1, g.buf);     /* as if a table draw had just used word 0 of call 0 */
        g.have = 2; g.after_table = 1;
        out[i] = hg_draw(&t, pop, good, sample, &g, ft);
    }
    free(t.G); free(t.LF); free(ft);
    return 0;
}

/* the tables the lock-step streams of a 3 x 2 / 2 x 3 row score: (x0, z) pairs in scoring order, 2 * n_shuffles words.
 * Returns 0, or -2 when the row does not take the lock-step path */
int lgo_lockstep_tables(const uint32_t T[9], uint64_t seed, uint32_t n_shuffles, uint32_t* rec)
{
    perm_tables t;
    first_table ft;
    uint32_t R[3], C[3], N = 0, nzr[3], nzc[3], nr = 0, nc = 0, a;
    for (a = 0; a < 3; ++a) {
        R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2];
        C[a] = T[a] + T[3 + a] + T[6 + a];
        N += R[a];
    }
    for (a = 0; a < 3; ++a) { if (R[a]) nzr[nr++] = a; if (C[a]) nzc[nc++] = a; }
    if (nr * nc != 6) return -2;
    if (tables_init(&t, N)) return -1;
    first_table_build(&t, N, R[nzr[0]], C[nzc[0]], &ft);
    if (ft.valid && N - R[nzr[0]] > 1 && N - C[nzc[0]] > 1)
        hrua_width_bound(&ft, N, (int)nr, R[nzr[0]], R[nzr[1]], C[nzc[0]], C[nzc[1]]);
    if (!simple_row(&ft, N, nr, nc, R[nzr[0]], C[nzc[0]], C[nzc[1]])) { free(t.G); free(t.LF); return -2; }
    (void)perm_lockstep(&t, &ft, N, nr == 3, R[nzr[0]], R[nzr[1]], nr == 3 ? R[nzr[2]] : 0, C[nzc[0]], C[nzc[1]],
                        stat9(&t, T), 3u, 4u, n_shuffles, (uint32_t)seed, (uint32_t)(seed >> 32), rec);
    free(t.G); free(t.LF);
    return 0;
}

int lgo_binom_draw_many(uint32_t n, uint64_t thr, uint64_t seed, uint32_t count, uint32_t* out)
{
    perm_tables t;
    uint32_t i;
    if (tables_init(&t, n)) return -1;
    for (i = 0; i < count; ++i) out[i] = binom_draw(&t, n, thr, i, 7u, (uint32_t)seed, (uint32_t)(seed >> 32));
    free(t.G); free(t.LF);
    return 0;
}

/* the threshold table of the first draw: returns the window length (0: no table), klo and thr[] */
int lgo_first_table(uint32_t pop, uint32_t good, uint32_t sample, uint32_t* klo, uint32_t* thr)
{
    perm_tables t;
    first_table ft;
    int n = 0;
    if (tables_init(&t, pop)) return -1;
    first_table_build(&t, pop, good, sample, &ft);
    if (ft.valid) {
        n = (int)ft.n;
        *klo = ft.klo;
        memcpy(thr, ft.thr, sizeof(uint32_t) * ft.n);
    }
    free(t.G); free(t.LF);
    return n;
}

int lgo_perm_ptail(const uint32_t T[9], double* ptail)
{
    perm_tables t;
    uint32_t n = 0;
    int k;
    for (k = 0; k < 9; ++k) n += T[k];
    if (tables_init(&t, n)) return -1;
    (void)perm_one(&t, T, 0, 1, 0, 0, ptail);
    free(t.G); free(t.LF);
    return 0;
}

int lgo_hg_draw_many(uint32_t pop, uint32_t good, uint32_t sample, uint64_t seed, uint32_t n, uint32_t* out)
{
    return lgo_hg_draw_many2(pop, good, sample, seed, n, out, 0);
}

/* test hook: the enumerated mass of a table larger than 2 x 2 (units of 2^-62) and its number of candidate tables;
 * returns 1 when the row is enumerated at this n_shuffles, 0 when it keeps the Monte-Carlo path, -1 on error */
int lgo_perm_enum_mass(const uint32_t T[9], uint32_t n_shuffles, uint64_t* mass, uint64_t* n_tables)
{
    perm_tables t;
    uint32_t R[3], C[3], N = 0, a, fa[2], fb[2], la, lb, radix[4];
    int ok;
    for (a = 0; a < 3; ++a) { R[a] = T[3 * a] + T[3 * a + 1] + T[3 * a + 2]; C[a] = T[a] + T[3 + a] + T[6 + a]; N += R[a]; }
    if (tables_init(&t, N)) return -1;
    *mass = 0; *n_tables = 0;
    ok = enum_plan(R, C, n_shuffles, fa, fb, &la, &lb, radix, n_tables);
    if (ok) *mass = enum_mass(&t, R, C, N, stat9(&t, T), fa, fb, la, lb, radix, *n_tables);
    free(t.G); free(t.LF);
    return ok;
}

/* test hook: the six-cell path of a table at this n_shuffles.  Returns 1 when the row takes it (0: Monte-Carlo, -1: error);
 * out[0] = mass of {S < S_obs} in units of 2^-62 summed over the AREA (six_inside: the independent check), out[1] = its
 * lattice points, out[2] = chords examined, out[3] = the zero test fired, out[4] = thr, out[5] = the same mass by the
 * perimeter walk (six_inside_walk: what thr is made of) */
int lgo_perm_six(const uint32_t T[9], uint32_t n_shuffles, uint64_t out[6])
{
    perm_tables t;
    six_row sx;
    uint32_t n = 0;
    int k, ok;
    for (k = 0; k < 9; ++k) n += T[k];
    if (tables_init(&t, n)) return -1;
    out[0] = out[1] = out[2] = out[3] = out[4] = out[5] = 0;
    ok = six_plan(&t, T, n_shuffles, &sx);
    if (ok) {
        out[3] = (uint64_t)sx.zero;
        if (!sx.zero) {
            out[0] = six_inside(&t, &sx, &out[1]);      /* the area sum: the independent check of the perimeter walk */
            out[2] = (uint64_t)(sx.zhi - sx.zlo - 1 > 0 ? sx.zhi - sx.zlo - 1 : 0);
            out[5] = six_inside_walk(&t, &sx);
        }
        out[4] = six_thr(&t, &sx, NULL);
    }
    free(t.G); free(t.LF);
    return ok;
}

/* test hook: lattice points per shuffle a six-cell row may cost (0: the path is off; the GPU library reads the same from
 * LGMI_PERM_SIX_PTS).  Returns the previous value. */
uint32_t lgo_set_six_pts(uint32_t v)
{
    const uint32_t old = SIX_PTS;
    SIX_PTS = v;
    return old;
}

/* test hook: the largest number of candidate tables a row is enumerated at (0: every larger-than-2x2 row keeps the
 * Monte-Carlo path); the GPU library reads the same from LGMI_PERM_ENUM_MAX.  Returns the previous value. */
uint32_t lgo_set_enum_max(uint32_t v)
{
    const uint32_t old = ENUM_MAX;
    ENUM_MAX = v;
    return old;
}
