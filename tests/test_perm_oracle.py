"""Statistical validation of the permutation-test SPECIFICATION (oracle/lgmi_perm_oracle.c).
The reference has no permutation test (parity unpinned), so the specification itself is
checked here against scipy's hypergeometric distribution and brute-force enumeration.
CPU only."""
import ctypes as C
import itertools

import numpy as np
import pytest
from scipy import stats

from oracle import c_oracle

u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)


@pytest.fixture(scope='module')
def lib():
    lib = c_oracle.load()
    lib.lgo_hg_draw_many.restype = C.c_int
    lib.lgo_hg_draw_many.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, u32p]
    lib.lgo_hg_draw_many2.restype = C.c_int
    lib.lgo_hg_draw_many2.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, u32p, C.c_int]
    lib.lgo_first_table.restype = C.c_int
    lib.lgo_first_table.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    lib.lgo_hg_draw_wide.restype = C.c_int
    lib.lgo_hg_draw_wide.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, u32p, C.c_double]
    lib.lgo_binom_draw_many.restype = C.c_int
    lib.lgo_binom_draw_many.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, u32p]
    lib.lgo_perm_ptail.restype = C.c_int
    lib.lgo_perm_ptail.argtypes = [u32p, f64p]
    lib.lgo_perm_six.restype = C.c_int
    lib.lgo_perm_six.argtypes = [u32p, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.lgo_set_six_pts.restype = C.c_uint32
    lib.lgo_set_six_pts.argtypes = [C.c_uint32]
    return lib


HG_CASES = [  # (pop, good, sample): urn path, HRUA path, complements, good > bad, extremes
    (20, 7, 5), (20, 15, 5), (20, 7, 16), (1000, 3, 9), (1000, 997, 9), (1000, 500, 995),
    (50, 25, 25), (300, 40, 150), (300, 260, 150), (300, 150, 20), (300, 150, 280),
    (5000, 2500, 2500), (5000, 100, 4000), (5000, 4900, 1000), (160000, 80000, 48000),
    (160000, 8000, 150000), (2000, 1990, 1000), (64, 10, 10), (64, 54, 54), (21, 11, 10), (21, 11, 11),
]


def _draws(lib, pop, good, sample, n, use_table):
    out = np.zeros(n, np.uint32)
    rc = lib.lgo_hg_draw_many2(pop, good, sample, 1234 + pop, n, out.ctypes.data_as(u32p), int(use_table))
    return rc, out


# first-draw threshold table: windows of a few to ~1500 entries, clipped at either support end, and too wide
TABLE_CASES = [(50, 25, 25), (300, 40, 150), (300, 260, 150), (5000, 2500, 2500), (5000, 100, 4000),
               (5000, 4900, 1000), (160000, 80000, 48000), (200000, 100000, 100000), (200000, 66000, 120000),
               (160000, 8000, 150000), (2000, 1990, 1000), (64, 54, 54), (21, 11, 10)]


@pytest.mark.parametrize('pop,good,sample', TABLE_CASES)
def test_first_draw_thresholds_are_the_scaled_cdf(lib, pop, good, sample):
    thr = np.zeros(4096, np.uint32)
    klo = C.c_uint32(0)
    n = lib.lgo_first_table(pop, good, sample, C.byref(klo), thr.ctypes.data_as(u32p))
    assert 0 < n <= 2032
    ks = klo.value + np.arange(n)
    lo, hi = max(0, sample + good - pop), min(good, sample)
    assert ks[0] >= lo and ks[-1] <= hi
    cdf = stats.hypergeom.cdf(ks, pop, good, sample)
    below = stats.hypergeom.cdf(klo.value - 1, pop, good, sample) if klo.value > lo else 0.0
    # the window drops < 1e-10 of the mass on either side; inside it thr = floor(2^32 * windowed CDF) +- 2
    assert below < 1e-10 and 1.0 - cdf[-1] < 1e-10
    want = (cdf - below) * 2.0 ** 32
    assert np.all(np.diff(thr[:n].astype(np.int64)) >= 0)
    assert np.max(np.abs(thr[:n].astype(np.float64) - np.minimum(want, 2.0 ** 32 - 1))) < 3.0 + 2.0 ** 32 * 1e-11


def test_first_draw_table_limits(lib):
    thr = np.zeros(4096, np.uint32)
    klo = C.c_uint32(0)
    for pop, good, sample in [(1000, 3, 9), (20, 7, 5), (400000, 200000, 200000), (1000, 0, 10), (1000, 1000, 10)]:
        # urn-sized, trivially determined, or wider than FIRST_MAX: no table, the rejection path stays
        assert lib.lgo_first_table(pop, good, sample, C.byref(klo), thr.ctypes.data_as(u32p)) == 0


@pytest.mark.parametrize('use_table', [False, True])
@pytest.mark.parametrize('pop,good,sample', HG_CASES)
def test_hypergeometric_sampler_matches_scipy(lib, pop, good, sample, use_table):
    n = 40000
    rc, out = _draws(lib, pop, good, sample, n, use_table)
    if use_table and rc == -2:
        pytest.skip('no threshold table for these parameters')
    assert rc == 0
    lo, hi = max(0, sample + good - pop), min(good, sample)
    assert out.min() >= lo and out.max() <= hi
    ks = np.arange(lo, hi + 1)
    pmf = stats.hypergeom.pmf(ks, pop, good, sample)
    obs = np.bincount(out - lo, minlength=len(ks)).astype(float)
    exp = pmf * n
    # pool the sparse tails so that every cell expects >= 8 draws
    order = np.argsort(-exp)
    keep = exp[order] >= 8
    o = np.append(obs[order][keep], obs[order][~keep].sum())
    e = np.append(exp[order][keep], exp[order][~keep].sum())
    if e[-1] < 1e-9:
        o, e = o[:-1], e[:-1]
    chi2 = ((o - e) ** 2 / e).sum()
    p = stats.chi2.sf(chi2, len(e) - 1)
    assert p > 1e-4, 'chi2 %.1f over %d cells, p=%.2e' % (chi2, len(e), p)
    mean = good * sample / pop
    assert abs(out.mean() - mean) < 5 * np.sqrt(max(stats.hypergeom.var(pop, good, sample), 1e-9) / n) + 1e-9


# (n, p): inversion branch (n p < 10), BTRS branch, both sides of the switch, p > 1/2 (mirrored), tiny and huge n
BINOM_CASES = [(1000, 0.5), (1000, 0.013), (1000, 0.0099), (1000, 0.0101), (1000, 0.3), (1000, 0.97), (1000, 0.9995),
               (1000, 1e-5), (64, 0.4), (20, 0.5), (50, 0.19), (50, 0.21), (100000, 0.2), (10000000, 0.5), (3, 0.5), (1, 0.3)]


@pytest.mark.parametrize('n,p', BINOM_CASES)
def test_binomial_sampler_matches_scipy(lib, n, p):
    """the 2 x 2 path draws the exceed count as ONE Binomial(n_shuffles, P_tail) variate"""
    cnt = 100000
    thr = int(p * 2 ** 32)
    out = np.zeros(cnt, np.uint32)
    assert lib.lgo_binom_draw_many(n, thr, 5, cnt, out.ctypes.data_as(u32p)) == 0
    pp = thr / 2 ** 32
    assert out.max() <= n
    lo, hi = int(out.min()), int(out.max())
    ks = np.arange(max(0, lo - 10), min(n, hi + 10) + 1)
    obs = np.bincount(out - ks[0], minlength=len(ks)).astype(float)
    exp = stats.binom.pmf(ks, n, pp) * cnt
    order = np.argsort(-exp)
    keep = exp[order] >= 8
    o = np.append(obs[order][keep], obs[order][~keep].sum())
    e = np.append(exp[order][keep], cnt - exp[order][keep].sum())
    if e[-1] < 1e-6:
        assert o[-1] == 0
        o, e = o[:-1], e[:-1]
    if len(e) > 1:
        chi2 = ((o - e) ** 2 / e).sum()
        pv = stats.chi2.sf(chi2, len(e) - 1)
        assert pv > 1e-4, 'chi2 %.1f over %d cells, p=%.2e' % (chi2, len(e), pv)
    assert abs(out.mean() - n * pp) < 5 * np.sqrt(n * pp * (1 - pp) / cnt) + 1e-9


def test_binomial_sampler_edges(lib):
    out = np.zeros(100, np.uint32)
    assert lib.lgo_binom_draw_many(1000, 0, 1, 100, out.ctypes.data_as(u32p)) == 0 and (out == 0).all()
    assert lib.lgo_binom_draw_many(1000, 2 ** 32, 1, 100, out.ctypes.data_as(u32p)) == 0 and (out == 1000).all()
    assert lib.lgo_binom_draw_many(0, 2 ** 31, 1, 100, out.ctypes.data_as(u32p)) == 0 and (out == 0).all()
    # thr = 1: p = 2^-32 -> essentially always 0; thr = 2^32 - 1 -> essentially always n
    assert lib.lgo_binom_draw_many(1000, 1, 1, 100, out.ctypes.data_as(u32p)) == 0 and (out == 0).all()
    assert lib.lgo_binom_draw_many(1000, 2 ** 32 - 1, 1, 100, out.ctypes.data_as(u32p)) == 0 and (out == 1000).all()


@pytest.mark.parametrize('factor', [1.0, 1.02, 1.5])
@pytest.mark.parametrize('pop,good,sample', [(300, 40, 150), (5000, 2500, 2500), (160000, 80000, 48000), (160000, 8000, 150000)])
def test_hrua_with_a_wider_hat_is_still_exact(lib, pop, good, sample, factor):
    """the draw that follows a table draw in a 3 x 2 row uses one hat width per row, the largest over the first
    draw's window: any width >= Stadlober's keeps the sampler exact (only the acceptance rate drops)"""
    n = 40000
    out = np.zeros(n, np.uint32)
    assert lib.lgo_hg_draw_wide(pop, good, sample, 91, n, out.ctypes.data_as(u32p), factor) == 0
    lo, hi = max(0, sample + good - pop), min(good, sample)
    assert out.min() >= lo and out.max() <= hi
    ks = np.arange(lo, hi + 1)
    obs = np.bincount(out - lo, minlength=len(ks)).astype(float)
    exp = stats.hypergeom.pmf(ks, pop, good, sample) * n
    order = np.argsort(-exp)
    keep = exp[order] >= 8
    o = np.append(obs[order][keep], obs[order][~keep].sum())
    e = np.append(exp[order][keep], n - exp[order][keep].sum())
    chi2 = ((o - e) ** 2 / e).sum()
    assert stats.chi2.sf(chi2, len(e) - 1) > 1e-4


def stat_of(table):
    t = np.asarray(table, float).ravel()
    return float(np.sum(np.where(t > 0, t * np.log(np.where(t > 0, t, 1.0)), 0.0)))


def brute_ptail_2x2(T):
    """exact P(S >= S_obs) band for a table with exactly 2 non-empty rows and columns"""
    T = np.asarray(T).reshape(3, 3)
    rows = [a for a in range(3) if T[a].sum()]
    cols = [b for b in range(3) if T[:, b].sum()]
    N, K, n = int(T.sum()), int(T[rows[1]].sum()), int(T[:, cols[1]].sum())
    kobs = int(T[rows[1], cols[1]])
    ks = np.arange(max(0, K + n - N), min(K, n) + 1)
    pmf = stats.hypergeom.pmf(ks, N, K, n)
    s = np.array([stat_of([[N - K - n + k, n - k], [K - k, k]]) for k in ks])
    sobs = stat_of([[N - K - n + kobs, n - kobs], [K - kobs, kobs]])
    tol = 1e-11 * max(1.0, abs(sobs))
    return pmf[s > sobs + tol].sum(), pmf[s >= sobs - tol].sum()


@pytest.mark.parametrize('seed', range(30))
def test_exact_tail_probability_2x2(lib, seed):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([6, 13, 40, 200, 3000, 60000]))
    a = rng.multinomial(N, rng.dirichlet(np.ones(4) * (0.5 + 3 * rng.random())))
    r, c = sorted(rng.choice(3, 2, replace=False)), sorted(rng.choice(3, 2, replace=False))
    T = np.zeros((3, 3), np.uint32)
    T[np.ix_(r, c)] = a.reshape(2, 2)
    if (T.sum(axis=1) > 0).sum() < 2 or (T.sum(axis=0) > 0).sum() < 2:
        pytest.skip('degenerate draw')
    p = C.c_double()
    assert lib.lgo_perm_ptail(np.ascontiguousarray(T.ravel()).ctypes.data_as(u32p), C.byref(p)) == 0
    lo, hi = brute_ptail_2x2(T)
    assert lo - 1e-9 <= p.value <= hi + 1e-9, (p.value, lo, hi)


def brute_band_fast(N, K, n, kobs):
    """brute_ptail_2x2 vectorised and restricted to mode +- 14 sigma (what lies beyond weighs < 1e-40)"""
    lo, hi = max(0, K + n - N), min(K, n)
    sd = np.sqrt(stats.hypergeom.var(N, K, n))
    ks = np.arange(max(lo, int(n * K / N - 14 * sd) - 2), min(hi, int(n * K / N + 14 * sd) + 2) + 1)
    ks = np.union1d(ks, [kobs])
    pmf = stats.hypergeom.pmf(ks, N, K, n)

    def xlogx(t):
        t = t.astype(float)
        return np.where(t > 0, t * np.log(np.where(t > 0, t, 1.0)), 0.0)
    st = xlogx(N - K - n + ks) + xlogx(n - ks) + xlogx(K - ks) + xlogx(ks)
    sobs = st[np.searchsorted(ks, kobs)]
    tol = 1e-11 * max(1.0, abs(sobs))
    return pmf[st > sobs + tol].sum(), pmf[st >= sobs - tol].sum()


@pytest.mark.parametrize('N', [500, 20000, 200000, 3000000])
def test_exact_tail_near_independence_and_moderate_tails(lib, N):
    """tables drawn around the null (centre form: 1 - mass of the less extreme values, up to ~7 sigma wide) and
    further out (tail form: two clipped runs), at read counts up to the millions"""
    rng = np.random.default_rng(N)
    for _ in range(12):
        K = int(rng.integers(N // 20, N - N // 20))
        n = int(rng.integers(N // 20, N - N // 20))
        sd = np.sqrt(stats.hypergeom.var(N, K, n))
        lo, hi = max(0, K + n - N), min(K, n)
        for z in (0.0, 0.3, -1.0, 2.2, -3.4, 3.6, -4.5, 6.0, -9.0):
            k = int(np.clip(round(n * K / N + z * sd), lo, hi))
            T = np.zeros((3, 3), np.uint32)
            T[1, 1], T[1, 2], T[2, 1], T[2, 2] = N - K - n + k, n - k, K - k, k
            pv = C.c_double()
            assert lib.lgo_perm_ptail(np.ascontiguousarray(T.ravel()).ctypes.data_as(u32p), C.byref(pv)) == 0
            blo, bhi = brute_band_fast(N, K, n, k)
            # a pmf from differences of log-factorials ~ 4e7 is good to ~1e-8 at millions of reads (the table's
            # libm lgamma; measured against a 60-digit Stirling sum), and the centre form inherits that as an
            # ABSOLUTE error of its complement
            tol = 1e-9 if N <= 200000 else 5e-8
            assert blo - tol <= pv.value <= bhi + tol, (N, K, n, k, pv.value, blo, bhi)
            if pv.value > 1e-6 and N <= 200000:
                assert blo * (1 - 1e-6) - 1e-12 <= pv.value <= bhi * (1 + 1e-6) + 1e-12


def run_perm(tables, n_shuffles, seed=7):
    lib = c_oracle.load()
    t = np.ascontiguousarray(np.asarray(tables, np.uint32).reshape(-1, 9))
    n = len(t)
    ri = np.arange(n, dtype=np.uint32)
    rj = ri + np.uint32(n)
    p = np.zeros(n)
    ex = np.zeros(n, np.uint32)
    rc = lib.lgo_perm_rows(n, ri.ctypes.data_as(u32p), rj.ctypes.data_as(u32p), t.ctypes.data_as(u32p),
                           n_shuffles, seed, p.ctypes.data_as(f64p), ex.ctypes.data_as(u32p), 0)
    assert rc == 0
    return p, ex


def test_degenerate_tables_have_p_one():
    p, ex = run_perm([[0, 0, 0, 0, 5, 9, 0, 0, 0], [0, 0, 0, 0, 4, 0, 0, 7, 0]], 100)
    assert (ex == 100).all() and (p == 1.0).all()


def test_monte_carlo_2x2_agrees_with_exact_tail(lib):
    tables = [[0, 0, 0, 0, 30, 10, 0, 12, 28], [0, 0, 0, 0, 5, 4, 0, 3, 6], [0, 0, 0, 0, 500, 480, 0, 470, 520],
              [7, 0, 3, 0, 0, 0, 2, 0, 9]]
    S = 40000
    p, ex = run_perm(tables, S)
    for T, e in zip(tables, ex):
        pt = C.c_double()
        lib.lgo_perm_ptail(np.asarray(T, np.uint32).ctypes.data_as(u32p), C.byref(pt))
        sd = np.sqrt(max(pt.value * (1 - pt.value), 1e-12) / S)
        assert abs(e / S - pt.value) < 5 * sd + 1e-9


def exact_p_general(T):
    """brute-force P(S >= S_obs) over all tables with the margins of T (small N only)"""
    T = np.asarray(T).reshape(3, 3)
    R, Cm, N = T.sum(axis=1), T.sum(axis=0), int(T.sum())
    sobs = stat_of(T)
    from math import lgamma
    const = sum(lgamma(x + 1) for x in R) + sum(lgamma(x + 1) for x in Cm) - lgamma(N + 1)
    tot = tail = 0.0
    for a, b, c, d in itertools.product(range(R[0] + 1), range(R[0] + 1), range(R[1] + 1), range(R[1] + 1)):
        t = np.array([[a, b, R[0] - a - b], [c, d, R[1] - c - d], [0, 0, 0]])
        t[2] = Cm - t[0] - t[1]
        if (t < 0).any():
            continue
        pr = np.exp(const - sum(lgamma(x + 1) for x in t.ravel()))
        tot += pr
        if stat_of(t) >= sobs - 1e-11 * max(1, abs(sobs)):
            tail += pr
    assert abs(tot - 1) < 1e-9
    return tail


@pytest.mark.parametrize('T', [[3, 2, 1, 1, 6, 2, 2, 1, 7], [4, 0, 2, 1, 5, 0, 0, 2, 6], [0, 0, 0, 2, 6, 3, 5, 1, 4],
                               [2, 0, 5, 1, 0, 8, 6, 0, 3], [10, 3, 2, 2, 9, 4, 1, 3, 12]])
def test_monte_carlo_general_tables_agree_with_enumeration(T):
    S = 30000
    p, ex = run_perm([T], S)
    exact = exact_p_general(T)
    sd = np.sqrt(max(exact * (1 - exact), 1e-12) / S)
    assert abs(ex[0] / S - exact) < 5 * sd + 2e-4, (ex[0] / S, exact)


def test_streams_are_keyed_by_seed_and_pair():
    T = [[0, 0, 0, 0, 30, 14, 0, 12, 28]] * 4
    p1, e1 = run_perm(T, 999, seed=1)
    p2, e2 = run_perm(T, 999, seed=2)
    p3, e3 = run_perm(T, 999, seed=1)
    assert (e1 == e3).all() and (e1 != e2).any() and len(set(e1.tolist())) > 1
    assert np.allclose(p1, (1 + e1) / 1000.0)


# ---------------------------------------------------------------- lock-step streams of 3 x 2 / 2 x 3 rows (round 3)
LOCKSTEP_TABLES = [
    # 3 x 2 (a tri-allelic site against a bi-allelic one): rows = classes 0, 1, 2 of site i
    [12, 30, 0, 60, 45, 0, 150, 140, 0],
    [0, 0, 0, 40, 90, 35, 120, 60, 55],          # 2 x 3
    [25, 0, 20, 300, 0, 310, 800, 0, 700],
    [0, 0, 0, 900, 400, 150, 300, 700, 160],
    [0, 60, 70, 0, 500, 450, 0, 2000, 2100],
]


def _margins(T):
    T = np.asarray(T).reshape(3, 3)
    R, Cm = T.sum(axis=1), T.sum(axis=0)
    return T, [int(x) for x in R if x], [int(x) for x in Cm if x]


def _lockstep_pmf(T):
    """exact joint pmf of (x0, z) of a six-cell table, with x0 and z as perm_lockstep defines them, and the statistic"""
    from scipy.special import gammaln
    T, R, Cm = _margins(T)
    N = sum(R)
    out = {}
    if len(R) == 3:          # x0 = T00, z = T10; column 0 holds (x0, z, C0 - x0 - z)
        R0, R1, R2 = R
        C0, C1 = Cm
        const = sum(gammaln(np.array(R) + 1.0)) + sum(gammaln(np.array(Cm) + 1.0)) - gammaln(N + 1.0)
        for x0 in range(0, min(R0, C0) + 1):
            for z in range(0, min(R1, C0 - x0) + 1):
                x2 = C0 - x0 - z
                if x2 > R2:
                    continue
                cells = np.array([x0, z, x2, R0 - x0, R1 - z, R2 - x2], float)
                out[(x0, z)] = (np.exp(const - gammaln(cells + 1.0).sum()), cells)
    else:                    # x0 = T00, z = T01; row 0 holds (x0, z, R0 - x0 - z)
        R0, R1 = R
        C0, C1, C2 = Cm
        const = sum(gammaln(np.array(R) + 1.0)) + sum(gammaln(np.array(Cm) + 1.0)) - gammaln(N + 1.0)
        for x0 in range(0, min(R0, C0) + 1):
            for z in range(0, min(C1, R0 - x0) + 1):
                t02 = R0 - x0 - z
                if t02 > C2 or C0 - x0 > R1 or C1 - z > R1 - (C0 - x0):
                    continue
                cells = np.array([x0, z, t02, C0 - x0, C1 - z, C2 - t02], float)
                out[(x0, z)] = (np.exp(const - gammaln(cells + 1.0).sum()), cells)
    return out


def _lockstep_rec(lib, T, S, seed=11):
    lib.lgo_lockstep_tables.restype = C.c_int
    lib.lgo_lockstep_tables.argtypes = [u32p, C.c_uint64, C.c_uint32, u32p]
    rec = np.zeros(2 * S, np.uint32)
    rc = lib.lgo_lockstep_tables(np.asarray(T, np.uint32).ctypes.data_as(u32p), seed, S, rec.ctypes.data_as(u32p))
    return rc, rec.reshape(-1, 2)


@pytest.mark.parametrize('T', LOCKSTEP_TABLES)
def test_lockstep_tables_follow_the_null_distribution(lib, T):
    """the (x0, z) the 64 lock-step streams score are draws from the multivariate hypergeometric null: chi-square of
    the joint counts against the exact pmf (cells pooled until they expect >= 8), and of each margin"""
    S = 200000
    rc, rec = _lockstep_rec(lib, T, S)
    assert rc == 0, 'the table does not take the lock-step path'
    pmf = _lockstep_pmf(T)
    assert abs(sum(p for p, _ in pmf.values()) - 1.0) < 1e-9
    keys = sorted(pmf, key=lambda k: -pmf[k][0])
    index = {k: n for n, k in enumerate(keys)}
    obs = np.zeros(len(keys))
    for x0, z in rec:
        obs[index[(int(x0), int(z))]] += 1          # a KeyError here = a table outside the support
    exp = np.array([pmf[k][0] for k in keys]) * S
    # pool the tail of the probability-sorted cells into one bin of expectation >= 8
    cut = int(np.searchsorted(-exp, -8.0))
    o = np.append(obs[:cut], obs[cut:].sum())
    e = np.append(exp[:cut], exp[cut:].sum())
    chi2 = ((o - e) ** 2 / np.maximum(e, 1e-300)).sum()
    assert stats.chi2.sf(chi2, len(o) - 1) > 1e-4, (chi2, len(o))
    for axis in (0, 1):                            # the two margins separately (far fewer bins: sharper)
        vals = np.array([k[axis] for k in keys])
        lo, hi = vals.min(), vals.max()
        e1 = np.bincount(vals - lo, weights=exp, minlength=hi - lo + 1)
        o1 = np.bincount(rec[:, axis].astype(np.int64) - lo, minlength=hi - lo + 1).astype(float)
        keep = e1 >= 8
        o1 = np.append(o1[keep], o1[~keep].sum())
        e1 = np.append(e1[keep], e1[~keep].sum())
        ok = e1 > 0
        chi2 = ((o1[ok] - e1[ok]) ** 2 / e1[ok]).sum()
        assert stats.chi2.sf(chi2, ok.sum() - 1) > 1e-4, (axis, chi2)


@pytest.mark.parametrize('T', LOCKSTEP_TABLES[:4])
def test_lockstep_monte_carlo_p_agrees_with_enumeration(lib, T):
    S = 60000
    old = lib.lgo_set_six_pts(0)                    # (the six-cell path would take these rows: this is the sampling loop's test)
    try:
        p, ex = run_perm([T], S)
    finally:
        lib.lgo_set_six_pts(old)
    pmf = _lockstep_pmf(T)
    Tm = np.asarray(T, float).reshape(3, 3)
    g = lambda c: float((np.where(c > 0, c * np.log(np.maximum(c, 1)), 0.0)).sum())
    sobs = g(Tm.ravel())
    exact = sum(pr for pr, cells in pmf.values() if g(cells) >= sobs - 1e-9 * max(1.0, abs(sobs)))
    sd = np.sqrt(max(exact * (1 - exact), 1e-12) / S)
    assert abs(ex[0] / S - exact) < 5 * sd + 2e-4, (ex[0] / S, exact)
    # exactly S tables were scored: the recorded list is full, and a second run is identical
    rc, rec = _lockstep_rec(lib, T, 1000)
    rc2, rec2 = _lockstep_rec(lib, T, 1000)
    assert rc == 0 and (rec == rec2).all() and (rec[:, 0] > 0).any()


def test_lockstep_prefix_property(lib):
    """the tables scored with n_shuffles = S are the first S of those scored with a larger n_shuffles (the streams do
    not depend on n_shuffles), so exceed(S) is a prefix count"""
    T = LOCKSTEP_TABLES[0]
    _, a = _lockstep_rec(lib, T, 700)
    _, b = _lockstep_rec(lib, T, 2500)
    assert (b[:700] == a).all()


def exact_p_fast(T):
    """P(S >= S_obs) over all tables with the margins of T, like exact_p_general but as one numpy grid over the four
    cells of the two SMALLEST rows and columns (independent of the specification's code; any choice of free cells
    enumerates the same set of tables)"""
    from scipy.special import gammaln
    M = np.asarray(T, np.int64).reshape(3, 3)
    ro, co = np.argsort(M.sum(axis=1), kind='stable'), np.argsort(M.sum(axis=0), kind='stable')
    M = M[ro][:, co]                                   # largest row and column last: their cells are the dependent ones
    R, Cm, N = M.sum(axis=1), M.sum(axis=0), int(M.sum())
    g = [np.arange(min(R[a], Cm[b]) + 1, dtype=np.int64) for a in (0, 1) for b in (0, 1)]
    x00, x01, x10, x11 = np.meshgrid(*g, indexing='ij', sparse=True)
    cells = [x00, x01, R[0] - x00 - x01, x10, x11, R[1] - x10 - x11, Cm[0] - x00 - x10, Cm[1] - x01 - x11]
    cells.append(R[2] - cells[6] - cells[7])
    cells = [np.broadcast_to(c, np.broadcast_shapes(*[k.shape for k in cells])) for c in cells]
    ok = np.ones(cells[0].shape, bool)
    for c in cells:
        ok &= c >= 0
    cells = [c[ok].astype(np.float64) for c in cells]
    const = gammaln(R + 1.0).sum() + gammaln(Cm + 1.0).sum() - gammaln(N + 1.0)
    pr = np.exp(const - sum(gammaln(c + 1.0) for c in cells))
    st = sum(np.where(c > 0, c * np.log(np.maximum(c, 1.0)), 0.0) for c in cells)
    sobs = float(sum(v * np.log(v) for v in M.ravel() if v > 0))
    assert abs(pr.sum() - 1) < 1e-9
    return float(pr[st >= sobs - 1e-9 * max(1.0, abs(sobs))].sum())


# ---- small tables by enumeration (round 3: enum_plan / enum_mass)
def _enum(lib, T, n_shuffles=100000):
    lib.lgo_perm_enum_mass.restype = C.c_int
    lib.lgo_perm_enum_mass.argtypes = [u32p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    mass, nt = C.c_uint64(0), C.c_uint64(0)
    rc = lib.lgo_perm_enum_mass(np.asarray(T, np.uint32).ctypes.data_as(u32p), n_shuffles, C.byref(mass), C.byref(nt))
    return rc, mass.value * 2.0 ** -62, nt.value


ENUM_TABLES = [[3, 2, 1, 1, 6, 2, 2, 1, 7], [4, 0, 2, 1, 5, 0, 0, 2, 6], [0, 0, 0, 2, 6, 3, 5, 1, 4], [2, 0, 5, 1, 0, 8, 6, 0, 3],
               [10, 3, 2, 2, 9, 4, 1, 3, 12], [40, 3, 0, 5, 30, 0, 2, 1, 0], [0, 25, 4, 0, 3, 31, 0, 6, 2], [7, 7, 7, 7, 7, 7, 7, 7, 7],
               [120, 2, 0, 3, 90, 0, 1, 4, 0], [1, 0, 0, 0, 1, 0, 0, 0, 1]]


@pytest.mark.parametrize('T', ENUM_TABLES)
def test_enumerated_mass_is_the_exact_tail_probability(lib, T):
    rc, mass, nt = _enum(lib, T)
    M = np.asarray(T).reshape(3, 3)
    R, Cm = M.sum(axis=1), M.sum(axis=0)
    la, lb = int(np.argmax(R)), int(np.argmax(Cm))            # first largest margin
    want = int(np.prod([min(R[a], Cm[b]) + 1 for a in range(3) if a != la for b in range(3) if b != lb]))
    assert rc == (1 if want <= 4096 else 0)
    if not rc:
        return
    assert nt == want
    exact = exact_p_fast(T)
    assert abs(mass - exact) <= 1e-11 * max(exact, 1e-30) + nt * 2.0 ** -61, (mass, exact)


def test_enumeration_limits(lib):
    # too many candidate tables, or more than 4 n_shuffles of them: the row keeps the Monte-Carlo path
    assert _enum(lib, [300, 200, 100, 150, 250, 90, 80, 120, 310])[0] == 0
    T = [10, 3, 0, 2, 9, 0, 1, 3, 0]
    nt = _enum(lib, T)[2]
    assert _enum(lib, T, n_shuffles=(nt + 3) // 4)[0] == 1 and _enum(lib, T, n_shuffles=(nt + 3) // 4 - 1)[0] == 0
    # a 3 x 2 row of a few hundred reads with a rare third allele: a few hundred tables
    rc, mass, nt = _enum(lib, [180, 0, 150, 12, 0, 9, 3, 0, 2], 1000)
    assert rc == 1 and nt <= 4000 and 0 < mass <= 1


@pytest.mark.parametrize('T', ENUM_TABLES[:4] + ENUM_TABLES[5:7])
def test_monte_carlo_and_enumeration_paths_agree(lib, T):
    """the same rows through both paths of the specification: exceed / S of the Monte-Carlo path (enumeration switched
    off) and of the enumeration path (one binomial variate) both sit on the exact tail probability"""
    lib.lgo_set_enum_max.restype = C.c_uint32
    lib.lgo_set_enum_max.argtypes = [C.c_uint32]
    S = 40000
    exact = exact_p_fast(T)
    sd = np.sqrt(max(exact * (1 - exact), 1e-12) / S)
    old = lib.lgo_set_enum_max(0)
    try:
        _p, ex_mc = run_perm([T], S)
    finally:
        lib.lgo_set_enum_max(old)
    _p, ex_enum = run_perm([T], S)
    assert abs(ex_mc[0] / S - exact) < 5 * sd + 2e-4 and abs(ex_enum[0] / S - exact) < 5 * sd + 2e-4


# ---------------------------------------------------------------- six-cell tables: exact mass along the perimeter (round 4)
def _six(lib, T, n_shuffles=10 ** 6):
    """-> (takes the path, dict(area mass, points, chords, zero test, thr, walk mass))"""
    out = (C.c_uint64 * 6)()
    rc = lib.lgo_perm_six(np.asarray(T, np.uint32).ctypes.data_as(u32p), n_shuffles, out)
    return rc, {'area': out[0] * 2.0 ** -62, 'points': out[1], 'chords': out[2], 'zero': out[3], 'thr': out[4],
                'walk': out[5] * 2.0 ** -62}


def brute_ptail_six(T):
    """P(S >= S_obs) over every table with the margins of a 3 x 2 / 2 x 3 table, as one numpy grid over two free cells
    (independent of the specification's code: no chords, no recurrences)"""
    from scipy.special import gammaln
    M = np.asarray(T, np.int64).reshape(3, 3)
    R, Cm = M.sum(axis=1), M.sum(axis=0)
    A, B = (R, Cm[Cm > 0]) if (R > 0).sum() == 3 else (Cm, R[R > 0])
    N, B0 = int(A.sum()), int(B[0])
    z = np.arange(0, min(A[0], B0) + 1)[:, None]
    x = np.arange(0, min(A[1], B0) + 1)[None, :]
    a2 = B0 - z - x
    ok = (a2 >= 0) & (a2 <= A[2])
    cells = [np.where(ok, c, 0) for c in (z + 0 * x, x + 0 * z, a2, A[0] - z + 0 * x, A[1] - x + 0 * z, A[2] - a2)]
    const = gammaln(A + 1.0).sum() + gammaln(B + 1.0).sum() - gammaln(N + 1.0)
    with np.errstate(over='ignore'):
        pr = np.where(ok, np.exp(const - sum(gammaln(c + 1.0) for c in cells)), 0.0)
    st = sum(np.where(c > 0, c * np.log(np.maximum(c, 1)), 0.0) for c in [c.astype(float) for c in cells])
    sobs = float(sum(v * np.log(v) for v in M.ravel() if v > 0))
    assert abs(pr.sum() - 1) < 1e-9
    return float(pr[ok & (st >= sobs - 2e-13 * max(1.0, abs(sobs)))].sum())


def _random_six_table(rng, N, assoc):
    pa = rng.dirichlet([3, 3, 0.6])[rng.permutation(3)]
    c = rng.uniform(0.1, 0.9)
    la = rng.choice(3, N, p=pa)
    lb = (rng.random(N) < c + assoc * (la == 0)).astype(int)
    M = np.zeros((3, 3), int)
    shape = rng.integers(4)
    if shape < 2:                                    # 3 x 2, the empty column in either place
        np.add.at(M, (la, lb + shape), 1)
    else:                                            # 2 x 3, the empty row in either place
        np.add.at(M, (lb + shape - 2, la), 1)
    return M.ravel()


@pytest.mark.parametrize('seed', range(40))
def test_six_cell_mass_is_the_exact_tail_probability(lib, seed):
    """both evaluations of the mass of {S < S_obs} — the sum over the area and the walk along the perimeter, which is what
    the threshold is made of — against a brute-force grid over all tables, for 3 x 2 and 2 x 3 tables of 60 .. 8000 reads,
    null and associated"""
    rng = np.random.default_rng(100 + seed)
    N = int(rng.choice([60, 300, 2000, 8000]))
    T = _random_six_table(rng, N, [0.0, 0.0, 0.08, 0.15][seed % 4])
    M = T.reshape(3, 3)
    if (M.sum(axis=1) > 0).sum() * (M.sum(axis=0) > 0).sum() != 6:
        pytest.skip('a class came out empty')
    rc, o = _six(lib, T)
    assert rc == 1
    exact = brute_ptail_six(T)
    if o['zero']:
        assert exact < 2.0 ** -33 and o['thr'] == 0
        return
    assert abs((1 - o['walk']) - exact) < 2e-9 and abs((1 - o['area']) - exact) < 2e-9, (o, exact)
    assert abs(o['walk'] - o['area']) < 1e-10
    assert o['thr'] == (2 ** 62 - int(round(o['walk'] * 2.0 ** 62))) >> 30 or abs(o['thr'] * 2.0 ** -32 - exact) < 1e-8


def test_six_cell_walk_equals_area_at_two_hundred_thousand_reads(lib):
    """north-star sized tables (162,000 common reads, a 5 % third allele): the perimeter walk — ~100 affine maps chained
    through 16-chord scans — against the plain sum over the ~1e4 tables of the area"""
    rng = np.random.default_rng(1)
    N, worst, pts = 162000, 0.0, []
    for it in range(12):
        e = rng.uniform(0.05, 0.5)
        pa = np.array([0.95 * (1 - e), 0.95 * e, 0.05]) if it % 2 else np.array([0.475, 0.475, 0.05])
        la = rng.choice(3, N, p=pa)
        lb = (rng.random(N) < (0.5 if it % 2 else rng.uniform(0.05, 0.5))).astype(int)
        M = np.zeros((3, 3), int)
        np.add.at(M, (la, lb), 1)
        rc, o = _six(lib, M.ravel())
        assert rc == 1 and not o['zero']
        worst = max(worst, abs(o['walk'] - o['area']))
        pts.append(o['points'] / max(o['chords'], 1))
    assert worst < 1e-9, worst
    assert np.mean(pts) > 30            # the walk really skips work: tens of tables per chord are never visited


def test_six_cell_zero_test_and_gate(lib):
    # a tri-allelic het SNP against a linked het SNP: no table as extreme as the observed one weighs 2^-33
    rc, o = _six(lib, [4000, 30, 0, 40, 3900, 0, 210, 190, 0], 1000)
    assert rc == 1 and o['zero'] == 1 and o['thr'] == 0
    # the gate: chords + central chord length against SIX_PTS n_shuffles / 16 (SIX_PTS = 16: as many chords as shuffles)
    T = [12000, 11800, 0, 6100, 5900, 0, 1300, 1250, 0]
    rc, o = _six(lib, T, 10 ** 6)
    assert rc == 1 and o['chords'] > 20
    need = o['chords'] + 1
    assert _six(lib, T, need - 2)[0] == 0 and _six(lib, T, 40 * need)[0] == 1
    old = lib.lgo_set_six_pts(0)
    try:
        assert _six(lib, T, 10 ** 6)[0] == 0
    finally:
        lib.lgo_set_six_pts(old)
    # no table below S_obs (the observed table is the table of independence): inside = 0, thr = 2^32, p = 1
    rc, o = _six(lib, [50, 50, 0, 30, 30, 0, 20, 20, 0], 1000)
    assert rc == 1 and o['walk'] == 0.0 and o['thr'] == 2 ** 32
    # 3 x 3 tables never take the path
    assert _six(lib, [10, 3, 2, 2, 9, 4, 1, 3, 12])[0] == 0


@pytest.mark.parametrize('T', LOCKSTEP_TABLES)
def test_six_cell_path_and_sampling_path_agree(lib, T):
    """the same rows through both paths of the specification: exceed / S of the lock-step sampling loop (six-cell path
    switched off) and of the perimeter walk (one binomial variate) both sit on the exact tail probability"""
    S = 40000
    exact = brute_ptail_six(T)
    sd = np.sqrt(max(exact * (1 - exact), 1e-12) / S)
    old = lib.lgo_set_six_pts(0)
    try:
        _p, ex_mc = run_perm([T], S)
    finally:
        lib.lgo_set_six_pts(old)
    rc, o = _six(lib, T, S)
    assert rc == 1
    _p, ex_six = run_perm([T], S)
    assert abs(ex_mc[0] / S - exact) < 5 * sd + 2e-4 and abs(ex_six[0] / S - exact) < 5 * sd + 2e-4
    assert abs(o['thr'] * 2.0 ** -32 - exact) < 1e-8


def test_exact_p_entry_covers_enumerated_and_six_cell_tables():
    """what lgmi_params.exact_2x2 returns for larger tables (round 4): the enumerated mass for tables with few candidates,
    one minus the perimeter walk's inside mass for 3 x 2 / 2 x 3 tables, NaN where neither reaches — against brute force"""
    small = ENUM_TABLES[:4] + ENUM_TABLES[5:7]              # (ENUM_TABLES[4] has more than 4096 candidate tables)
    tables = small + LOCKSTEP_TABLES + [ENUM_TABLES[4], [300, 200, 100, 150, 250, 90, 80, 120, 310]]
    p = c_oracle.perm_rows_exact(np.array(tables))
    for T, got in zip(small, p[:6]):
        assert abs(got - exact_p_fast(T)) < 1e-10, T
    for T, got in zip(LOCKSTEP_TABLES, p[6:11]):
        assert abs(got - brute_ptail_six(T)) < 2e-9, T
    assert np.isnan(p[11]) and np.isnan(p[12])              # 3 x 3 tables beyond the enumeration: no exact form within reach


def test_exact_p_of_a_row_the_zero_test_decides_is_its_bound_never_zero(lib):
    """advice r4: a six-cell row whose tail the zero test bounds below 2^-33 used to come back with the exact p 0.0 — a
    value no permutation test can give (the observed table is in its own tail) and one that breaks -log10(p).  It is now
    the bound the test fired on: an UPPER bound of the tail mass (brute force below), positive, below e^-23.1; floored at
    the smallest normal double when even the bound underflows"""
    T = [400, 30, 0, 40, 390, 0, 30, 25, 0]                # a tri-allelic site against a linked one, small enough for brute force
    rc, o = _six(lib, T, 1000)
    assert rc == 1 and o['zero'] == 1
    p = c_oracle.perm_rows_exact(np.array([T]))[0]
    exact = brute_ptail_six(T)
    assert 0.0 < exact <= p < 9.4e-11, (exact, p)
    # strongly linked, deep: the bound itself underflows — the floor, not 0.0
    p = c_oracle.perm_rows_exact(np.array([[90000, 30, 0, 40, 89000, 0, 2100, 1900, 0]]))[0]
    assert p == 2.2250738585072014e-308


@pytest.mark.parametrize('T', [[0, 0, 0, 0, 30, 14, 0, 12, 28], [12, 30, 0, 60, 45, 0, 40, 28, 0], [0, 0, 0, 40, 35, 20, 25, 60, 18],
                               [10, 3, 2, 2, 9, 4, 1, 3, 12]], ids=['2x2', '3x2', '2x3', '3x3'])
def test_p_is_the_literal_label_shuffle_test_of_the_references_statistic(T):
    """the DEFINITION, literally and independently of every table-level shortcut of the specification: expand the table into
    the two label vectors the reference hands to scikit-learn (src/giremi/mutual_information.py:25-41), shuffle one of them
    with numpy, recompute sklearn.metrics.mutual_info_score, count MI_s >= MI_obs.  The specification's exceed / S (2 x 2:
    exact tail + binomial; 3 x 2 / 2 x 3: perimeter walk + binomial; 3 x 3: enumeration or sampled tables) must agree with
    that proportion within binomial noise — on both sides of the comparison"""
    from sklearn.metrics import mutual_info_score
    M = np.asarray(T).reshape(3, 3)
    a = np.repeat(np.arange(3), M.sum(axis=1))
    b = np.concatenate([np.repeat(np.arange(3), M[k]) for k in range(3)])
    obs = mutual_info_score(a, b)
    rng = np.random.default_rng(99)
    S_lit = 4000
    hits = sum(mutual_info_score(a, rng.permutation(b)) >= obs - 1e-12 for _ in range(S_lit))
    S = 40000
    _p, ex = run_perm([T], S)
    p_lit, p_spec = hits / S_lit, ex[0] / S
    sd = np.sqrt(p_lit * (1 - p_lit) / S_lit + p_spec * (1 - p_spec) / S + 1e-9)
    assert abs(p_lit - p_spec) < 4.5 * sd + 1e-3, (p_lit, p_spec)
