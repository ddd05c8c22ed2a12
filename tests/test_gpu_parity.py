"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden
vectors and against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): counts bit-exact, MI / mean MI within 1e-6."""
import os
import sys

import numpy as np
import pytest

from conftest import all_pair_cases, sites_to_mismatches
from util_synth import random_batch

pytestmark = pytest.mark.gpu

MI_TOL = 1e-6


@pytest.fixture(scope='module')
def engine():
    import lgmi
    eng = lgmi.Engine(0)
    yield eng
    eng.close()


def assert_same_as_oracle(res, ora, check_counts=True):
    assert res.n_rows == len(ora['row_i'])
    np.testing.assert_array_equal(res.row_i, ora['row_i'])
    np.testing.assert_array_equal(res.row_j, ora['row_j'])
    if check_counts:
        np.testing.assert_array_equal(res.row_counts, ora['row_counts'])        # bit-exact
    if res.n_rows:
        assert np.max(np.abs(res.row_mi - ora['row_mi'])) <= MI_TOL
        assert ((res.row_mi == 0.0) == (ora['row_mi'] == 0.0)).all()
    np.testing.assert_array_equal(res.site_n_pairs, ora['site_n_pairs'])
    m = ora['site_n_pairs'] > 0
    assert np.isnan(res.site_mean_mi[~m]).all()
    if m.any():
        assert np.max(np.abs(res.site_mean_mi[m] - ora['site_mean_mi'][m])) <= MI_TOL


# ---------------------------------------------------------------- reference golden vectors
@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_dropin_pair_mi_matches_reference(engine, case):
    import lgmi
    mm = sites_to_mismatches(case['sites'])
    rows = lgmi.mismatch_pair_mutual_info(mm, case['min_common'], engine=engine)
    assert [r[:4] for r in rows] == [r[:4] for r in case['rows']]
    for got, exp in zip(rows, case['rows']):
        assert abs(got[4] - exp[4]) <= MI_TOL
        assert isinstance(got[4], float)
        if exp[4] == 0.0:
            assert got[4] == 0.0
    # 3x3 tables through the batched entry, bit-exact against the reference's label vectors
    if len(mm) > 1:
        res = engine.run(lgmi.pack_blocks([mm]), min_common=case['min_common'], het_only=False, emit_counts=True)
        assert res.row_counts.reshape(-1, 9).tolist() == case['tables']


@pytest.mark.parametrize('case', all_pair_cases(), ids=lambda c: c['name'])
def test_dropin_mean_mi_matches_reference(engine, case):
    import lgmi
    for key, rows in (('mean_all', case['rows']),
                      ('mean_het', [r for r in case['rows'] if r[1] == 'het_snp' or r[3] == 'het_snp'])):
        got = lgmi.mean_mismatch_pair_mutual_info(rows, engine=engine)
        assert [g[0] for g in got] == [e[0] for e in case[key]]
        for g, e in zip(got, case[key]):
            assert abs(g[1] - e[1]) <= MI_TOL


def test_region_pair_mi_matches_reference_filter(engine):
    """mismatch.py:384-418: '+' rows then '-' rows, het_snp-involved pairs only, mean over kept rows"""
    import lgmi
    cases = {c['name']: c for c in all_pair_cases()}
    plus, minus = cases['random_03_P%s' % cases_suffix(cases, 'random_03')], cases['three_allele']
    mm = {'+': sites_to_mismatches(plus['sites']), '-': sites_to_mismatches(minus['sites'])}
    # both goldens were generated with their own min_common; use one the two share
    mc = 5
    from oracle import mi_oracle
    kept, means = mi_oracle.region_mi(mm, mc)
    records, mean_mi, pv = lgmi.region_pair_mi(mm, 'chrS', mc, engine=engine)
    exp = [['chrS', '+'] + r for r in kept['+']] + [['chrS', '-'] + r for r in kept['-']]
    assert pv is None
    assert [r[:6] for r in records] == [r[:6] for r in exp]
    for g, e in zip(records, exp):
        assert abs(g[6] - e[6]) <= MI_TOL
    for strand in '+-':
        em = dict(means[strand])
        assert set(em) == set(mean_mi[strand])
        for p in em:
            assert abs(em[p] - mean_mi[strand][p]) <= MI_TOL


def cases_suffix(cases, prefix):
    name = [n for n in cases if n.startswith(prefix)][0]
    return name[len(prefix) + 2:]


# ---------------------------------------------------------------- error behaviour of the reference
def test_zero_common_with_min_common_zero_raises_valueerror(engine):
    import lgmi
    mm = sites_to_mismatches([c for c in all_pair_cases() if c['name'] == 'below_min_common'][0]['sites'])
    mm[99999] = {'type': 'snp', 'depth': {'A': 5, 'G': 3}, 'nt': {'A': ['x1'], 'G': ['x2']}}
    with pytest.raises(ValueError, match='math domain error'):
        lgmi.mismatch_pair_mutual_info(mm, 0, engine=engine)


def test_fewer_than_two_alleles_raises_indexerror(engine):
    import lgmi
    mm = {1: {'type': 'mismatch', 'depth': {'A': 9}, 'nt': {'A': ['a', 'b']}},
          2: {'type': 'mismatch', 'depth': {'A': 9, 'C': 2}, 'nt': {'A': ['a'], 'C': ['b']}}}
    with pytest.raises(IndexError):
        lgmi.mismatch_pair_mutual_info(mm, 1, engine=engine)
    assert lgmi.mismatch_pair_mutual_info(mm, 5, engine=engine) == []   # never ranked: no pair passes :19


def test_region_block_raises_for_a_bad_site_in_a_pair_without_het_side(engine):
    """the reference ranks the alleles of every qualifying pair before its het filter (mutual_information.py:25-32,
    mismatch.py:392-396): a site with < 2 alleles in `depth` raises IndexError even when its only partner with enough
    common reads is not a het_snp"""
    import lgmi
    reads = ['r%d' % k for k in range(8)]
    mm = {'+': {10: {'type': 'mismatch', 'depth': {'A': 8}, 'nt': {'A': reads}},
                20: {'type': 'mismatch', 'depth': {'A': 5, 'C': 3}, 'nt': {'A': reads[:5], 'C': reads[5:]}},
                30: {'type': 'het_snp', 'depth': {'G': 2, 'T': 2}, 'nt': {'G': ['x1', 'x2'], 'T': ['x3', 'x4']}}},
          '-': {}}
    with pytest.raises(IndexError):
        lgmi.region_pair_mi(mm, 'chrS', 5, engine=engine)
    mm['+'][10]['nt'] = {'A': reads[:3]}                       # now no pair of site 10 has 5 common reads
    records, _mean, _p = lgmi.region_pair_mi(mm, 'chrS', 5, engine=engine)
    assert records == []


def test_empty_and_single_site(engine):
    import lgmi
    assert lgmi.mismatch_pair_mutual_info({}, 5, engine=engine) == []
    assert lgmi.mean_mismatch_pair_mutual_info([], engine=engine) == []
    one = {7: {'type': 'het_snp', 'depth': {'A': 3, 'G': 3}, 'nt': {'A': ['a', 'b', 'c'], 'G': ['d', 'e', 'f']}}}
    assert lgmi.mismatch_pair_mutual_info(one, 1, engine=engine) == []
    rec, means, _ = lgmi.region_pair_mi({'+': one, '-': {}}, 'c', 5, engine=engine)
    assert rec == [] and means == {'+': {}, '-': {}}


# ---------------------------------------------------------------- seeded random batches vs the C oracle
@pytest.mark.parametrize('seed', range(12))
@pytest.mark.parametrize('het_only', [True, False])
def test_random_batches_match_oracle(engine, seed, het_only):
    from oracle import c_oracle
    pb = random_batch(1000 + seed, n_blocks=1 + seed % 4, tri_frac=0.15 if seed % 3 else 0.0)
    mc = [1, 5, 6, 20][seed % 4]
    ora = c_oracle.run(pb, min_common=mc, het_only=het_only)
    res = engine.run(pb, min_common=mc, het_only=het_only, emit_counts=True)
    assert_same_as_oracle(res, ora)
    assert res.info['n_examined'] == ora['n_examined']


def test_many_small_blocks_and_empty_blocks(engine):
    from oracle import c_oracle
    from util_synth import pack_class_matrix, random_block
    rng = np.random.Generator(np.random.PCG64(5))
    blocks = []
    for k in range(60):
        P = int(rng.integers(0, 9))
        R = int(rng.integers(1, 130))
        if P == 0:
            blocks.append((np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros((0, R), np.int8)))
        else:
            blocks.append(random_block(rng, P, R, banded=False, tri_frac=0.2, het_frac=0.5))
    pb = pack_class_matrix(blocks)
    for het_only in (True, False):
        ora = c_oracle.run(pb, min_common=2, het_only=het_only)
        res = engine.run(pb, min_common=2, het_only=het_only, emit_counts=True)
        assert_same_as_oracle(res, ora)


def test_tile_edges_and_long_bands(engine):
    """sizes around the 64-column tile edge and the 8-word LDS stage"""
    from oracle import c_oracle
    from util_synth import pack_class_matrix, random_block
    rng = np.random.Generator(np.random.PCG64(11))
    for P, R in ((63, 64), (64, 65), (65, 511), (129, 513), (200, 1025)):
        pb = pack_class_matrix([random_block(rng, P, R, banded=(P % 2 == 1), tri_frac=0.1, het_frac=0.4)])
        ora = c_oracle.run(pb, min_common=3, het_only=True)
        res = engine.run(pb, min_common=3, het_only=True, emit_counts=True)
        assert_same_as_oracle(res, ora)


def test_no_het_sites_gives_no_rows(engine):
    from util_synth import pack_class_matrix, random_block
    rng = np.random.Generator(np.random.PCG64(3))
    pos, typ, cls = random_block(rng, 20, 100)
    typ[:] = 0
    res = engine.run(pack_class_matrix([(pos, typ, cls)]), min_common=1, het_only=True)
    assert res.n_rows == 0 and np.isnan(res.site_mean_mi).all()


# ---------------------------------------------------------------- device-side synthetic chromosome
def test_synth_dense_roundtrip_and_parity(engine):
    import lgmi
    from oracle import c_oracle
    spec = lgmi.default_synth_spec(150, 3000, seed=42)
    spec.tri_per_1024 = 200
    db = engine.synth_dense(spec)
    pb = db.download()
    assert pb.n_sites == 150 and int(pb.block_n_reads[0]) == 3000
    assert (pb.site_type[::5] == 2).all()
    # the downloaded batch, re-uploaded, is the same problem
    dr = engine.run_device(db, min_common=6, het_only=True, emit_counts=True)
    res = dr.fetch()
    ora = c_oracle.run(pb, min_common=6, het_only=True)
    assert_same_as_oracle(res, ora)
    res2 = engine.run(pb, min_common=6, het_only=True, emit_counts=True)
    np.testing.assert_array_equal(res.row_counts, res2.row_counts)
    np.testing.assert_array_equal(res.row_mi, res2.row_mi)
    # dropout ~10 %, so common reads ~ 0.81 R; het pairs are strongly linked
    n = res.row_counts.sum(axis=(1, 2))
    assert 0.75 * 3000 < n.mean() < 0.87 * 3000
    info = dr.info()
    assert info['n_rows'] == res.n_rows and info['ms_count'] > 0
    dr.free()
    db.free()


def test_run_is_deterministic(engine):
    pb = random_batch(77, n_blocks=2)
    a = engine.run(pb, min_common=3, het_only=True, emit_counts=True)
    b = engine.run(pb, min_common=3, het_only=True, emit_counts=True)
    for f in ('row_i', 'row_j', 'row_mi', 'row_counts', 'site_mean_mi', 'site_n_pairs'):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f))


# ---------------------------------------------------------------- permutation p-values (parity unpinned: vs the CPU specification)
def assert_perm_same(res, ora, n_shuffles):
    np.testing.assert_array_equal(res.row_i, ora['row_i'])
    np.testing.assert_array_equal(res.row_exceed, ora['row_exceed'])        # bit-exact
    np.testing.assert_array_equal(res.row_p, ora['row_p'])
    np.testing.assert_array_equal(res.row_p, (1.0 + res.row_exceed) / (n_shuffles + 1.0))


@pytest.mark.parametrize('seed', range(8))
def test_permutation_p_matches_cpu_specification(engine, seed):
    from oracle import c_oracle
    pb = random_batch(3000 + seed, n_blocks=1 + seed % 3, tri_frac=0.25 if seed % 2 else 0.0, R=(6, 900))
    S = [1, 7, 100, 257][seed % 4]
    ora = c_oracle.run(pb, min_common=[1, 5][seed % 2], het_only=True, n_shuffles=S, seed=99 + seed)
    res = engine.run(pb, min_common=[1, 5][seed % 2], het_only=True, n_shuffles=S, seed=99 + seed, emit_counts=True)
    assert_same_as_oracle(res, ora)
    assert_perm_same(res, ora, S)


@pytest.mark.parametrize('n_reads,S', [(40000, 64), (200000, 150), (420000, 70)])
def test_permutation_p_large_counts_and_tri_sites(engine, n_reads, S):
    """thousands of common reads: HRUA draws, exact-tail sums in both the centre and the tail form; the first
    draw of a general table comes from the threshold table (windows of ~660 and ~1460 entries) or, at 420k
    reads, from the rejection sampler because the window would not fit; S > 64 makes lanes take further
    shuffles from the shared counter"""
    import lgmi
    from oracle import c_oracle
    spec = lgmi.default_synth_spec(60 if n_reads == 40000 else 36, n_reads, seed=5)
    spec.tri_per_1024 = 250
    db = engine.synth_dense(spec)
    pb = db.download()
    res = engine.run_device(db, min_common=6, het_only=True, n_shuffles=S, seed=1234).fetch()
    ora = c_oracle.run(pb, min_common=6, het_only=True, n_shuffles=S, seed=1234)
    assert_perm_same(res, ora, S)
    # linked het pairs are significant, independent pairs are not systematically so
    t = pb.site_type
    hh = (t[res.row_i] == 2) & (t[res.row_j] == 2)
    assert hh.any() and (res.row_p[hh] == 1.0 / (S + 1)).all()
    assert np.median(res.row_p[~hh]) > 0.2
    db.free()


def test_exact_2x2_p_matches_cpu_specification(engine):
    """lgmi_params.exact_2x2: rows with at most 2 x 2 non-empty classes return the exact mass of the tables at least
    as extreme (bit-equal to the CPU specification, itself checked against brute-force enumeration in
    tests/test_perm_oracle.py); so do, since round 4, larger tables within reach of the enumeration or of the six-cell
    perimeter walk; the other larger tables keep the Monte-Carlo estimate, or NaN when n_shuffles == 0"""
    from lgmi._lib import EXCEED_EXACT
    from oracle import c_oracle
    pb = random_batch(8181, n_blocks=3, tri_frac=0.3, R=(6, 900))
    base = engine.run(pb, min_common=3, het_only=False, n_shuffles=60, seed=4, emit_counts=True)
    exact = c_oracle.perm_rows_exact(base.row_counts)
    small = ~np.isnan(exact)
    assert small.any() and (~small).any()
    c = base.row_counts.reshape(-1, 3, 3)
    shape = (c.sum(axis=2) > 0).sum(axis=1) * (c.sum(axis=1) > 0).sum(axis=1)
    assert (small & (shape == 6)).sum() > 20 and (small & (shape == 9)).any()      # exact p of 3 x 2 / 2 x 3 and of small 3 x 3 tables
    only = engine.run(pb, min_common=3, het_only=False, n_shuffles=0, exact_2x2=True, emit_counts=True)
    np.testing.assert_array_equal(only.row_counts, base.row_counts)
    np.testing.assert_array_equal(only.row_p, exact)                         # NaN-aware, bit for bit
    assert (only.row_exceed == EXCEED_EXACT).all()
    assert ((only.row_p[small] >= 0) & (only.row_p[small] <= 1)).all()
    both = engine.run(pb, min_common=3, het_only=False, n_shuffles=60, seed=4, exact_2x2=True, emit_counts=True)
    np.testing.assert_array_equal(both.row_p[small], exact[small])
    assert (both.row_exceed[small] == EXCEED_EXACT).all()
    np.testing.assert_array_equal(both.row_exceed[~small], base.row_exceed[~small])
    np.testing.assert_array_equal(both.row_p[~small], base.row_p[~small])
    # the Monte-Carlo estimate of the same rows scatters around the exact value like a binomial proportion
    z = (base.row_exceed[small] - 60 * exact[small]) / np.sqrt(60 * exact[small] * (1 - exact[small]) + 1e-9)
    assert abs(np.mean(z)) < 0.2 and np.mean(np.abs(z) < 3.5) > 0.99


def test_hardware_exp_stays_inside_the_guard(engine):
    """perm.hip le_exp(): the hardware's f32 exp decides an HRUA acceptance unless x2 is within 1e-4 of exp(t); that
    is the same decision as det_exp's as long as |__expf / det_exp - 1| < 1e-4.  Swept on the device over the
    whole range where the f32 value is a normal number, then adversarial x2 right at the guard's edges."""
    rng = np.random.default_rng(17)
    t = np.concatenate([np.linspace(-87.0, 0.0, 400_001), -rng.random(200_000) * 87.0, -rng.random(100_000) * 1e-3,
                        -np.exp(rng.uniform(-30, 4.4, 100_000))])
    t = t[t >= -87.0]
    _, _, e_hw, e_det = engine.selftest_le_exp(np.ones_like(t), t)
    rel = np.abs(e_hw / e_det - 1.0)
    assert rel.max() < 2.5e-5, rel.max()                                       # a quarter of the 1e-4 guard
    assert np.max(np.abs(e_det / np.exp(t) - 1.0)) < 1e-12                     # det_exp itself
    # below the normal range of f32 the hardware value may flush to 0: then exp(t) < 1.2e-38 < 2^-66 <= x2
    tl = -rng.uniform(87.0, 120.0, 50_000)
    fast, det, e_hw2, e_det2 = engine.selftest_le_exp(np.full_like(tl, 2.0 ** -66), tl)
    assert (e_det2 < 2.0 ** -66).all() and not fast.any() and not det.any()
    # decisions: x2 = exp(t) * (1 + d) for d on both sides of the guard and of the boundary itself
    ds = np.array([0.0, 1e-15, -1e-15, 2e-5, -2e-5, 9.0e-5, -9.0e-5, 0.99e-4, -0.99e-4, 1.01e-4, -1.01e-4, 3e-4, -3e-4,
                   0.3, -0.3])
    tt = np.repeat(-rng.random(60_000) * 45.0, len(ds))                       # x2 >= 2^-66 as in the callers
    x2 = np.exp(tt) * (1.0 + np.tile(ds, 60_000))
    keep = x2 >= 2.0 ** -66
    fast, det, _, _ = engine.selftest_le_exp(x2[keep], tt[keep])
    np.testing.assert_array_equal(fast, det)


def test_row_arrays_sized_exactly_or_by_their_bound(engine, monkeypatch):
    """lgmi_run_device sizes the row arrays by the examined-pair bound when that fits (no host wait between the two emit
    passes) and by one read-back of the row count otherwise (LGMI_EXACT_ROW_ALLOC forces it): same results either way,
    on a banded batch where far fewer rows are emitted than examined"""
    from lgmi.synth import banded_chromosome
    from oracle import c_oracle
    pb = banded_chromosome(3000, 12000, seed=5)
    ora = c_oracle.run(pb, min_common=6, het_only=True, n_shuffles=20, seed=8)
    a = engine.run(pb, min_common=6, het_only=True, n_shuffles=20, seed=8, emit_counts=True)
    monkeypatch.setenv('LGMI_EXACT_ROW_ALLOC', '1')
    b = engine.run(pb, min_common=6, het_only=True, n_shuffles=20, seed=8, emit_counts=True)
    assert a.n_rows < 0.5 * a.info['n_examined']
    for res in (a, b):
        assert_same_as_oracle(res, ora)
        assert_perm_same(res, ora, 20)


def test_permutation_seed_changes_draws_not_counts(engine):
    pb = random_batch(4242, n_blocks=1, P=(30, 30), R=(300, 300))
    a = engine.run(pb, min_common=5, n_shuffles=500, seed=1, emit_counts=True)
    b = engine.run(pb, min_common=5, n_shuffles=500, seed=2, emit_counts=True)
    np.testing.assert_array_equal(a.row_counts, b.row_counts)
    np.testing.assert_array_equal(a.row_mi, b.row_mi)
    assert (a.row_exceed != b.row_exceed).any()
    assert a.row_p.min() >= 1 / 501 and a.row_p.max() <= 1.0


# ---------------------------------------------------------------- RCCL gather (one rank is all a 1-GPU box allows)
def test_rccl_single_rank_gather(engine):
    pb = random_batch(555, n_blocks=2)
    uid = engine.comm_unique_id()
    assert len(uid) == 128
    engine.comm_init(uid, 0, 1)
    db = engine.upload(pb)
    dr = engine.run_device(db, min_common=3, het_only=True, n_shuffles=10, seed=3)
    res = dr.fetch()
    assert engine.comm_allgather_u64(res.n_rows) == [res.n_rows]
    g = engine.comm_gather_rows(dr, root=0)
    np.testing.assert_array_equal(g['row_i'], res.row_i)
    np.testing.assert_array_equal(g['row_j'], res.row_j)
    np.testing.assert_array_equal(g['row_mi'], res.row_mi)
    np.testing.assert_array_equal(g['row_p'], res.row_p)
    dr.free()
    db.free()


def test_banded_chromosome_matches_oracle(engine):
    """the long-read-like regime: several blocks, narrow bands, most tiles skipped"""
    from lgmi.synth import banded_chromosome
    from oracle import c_oracle
    pb = banded_chromosome(6000, 24000, seed=77)
    assert pb.n_blocks == 3 and pb.site_n_words.max() < 40
    ora = c_oracle.run(pb, min_common=6, het_only=True, n_shuffles=50, seed=5)
    res = engine.run(pb, min_common=6, het_only=True, n_shuffles=50, seed=5, emit_counts=True)
    assert_same_as_oracle(res, ora)
    assert_perm_same(res, ora, 50)
    assert res.info['n_examined'] == ora['n_examined'] and res.n_rows < 0.2 * ora['n_examined']


# ---------------------------------------------------------------- the reference's ECDF `mip` (stat.py:7-29)
def test_ecdf_matches_reference_vectors(engine):
    import lgmi
    from conftest import load_golden
    for c in load_golden('ecdf.json')['cases']:
        f = lgmi.ecdf(c['sample'], engine=engine)
        got = f(np.array(c['query']))
        np.testing.assert_array_equal(got, np.array(c['value']))      # the reference's linspace arithmetic, bit for bit
        assert f(c['query'][0]) == c['value'][0]                       # scalar call
    with pytest.raises(ZeroDivisionError):
        lgmi.ecdf([], engine=engine)


def test_mip_column(engine):
    import lgmi
    from oracle import mi_oracle
    rng = np.random.default_rng(1)
    mean = rng.random(500)
    mean[rng.random(500) < 0.2] = np.nan
    typ = np.where(rng.random(500) < 0.3, 'het_snp', 'mismatch')
    got = lgmi.mean_mi_to_mip(mean, typ, engine=engine)
    f = mi_oracle.ecdf_strict([v for v, t in zip(mean, typ) if t == 'het_snp' and not np.isnan(v)])
    for v, g in zip(mean, got):
        assert (np.isnan(v) and np.isnan(g)) or abs(f(v) - g) <= 1e-12
    assert np.isnan(lgmi.mean_mi_to_mip([np.nan, np.nan], np.array(['het_snp', 'snp']), engine=engine)).all()


# ---------------------------------------------------------------- the count kernels (VALU popcount / FP4 and int8 matrix cores)
@pytest.mark.parametrize('kernel', ['valu', 'mfma', 'mfma_i8'])
@pytest.mark.parametrize('seed', range(6))
def test_both_count_kernels_match_oracle(engine, monkeypatch, kernel, seed):
    from oracle import c_oracle
    from util_synth import pack_class_matrix, random_block
    monkeypatch.setenv('LGMI_COUNT_KERNEL', kernel)
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    shapes = [(130, 300), (257, 4100), (64, 64), (200, 1000), (129, 129), (40, 5000)]
    P, R = shapes[seed]
    blocks = [random_block(rng, P, R, banded=bool(seed % 2), tri_frac=0.15, het_frac=0.5),
              random_block(rng, 10 + seed, 70, tri_frac=0.2, het_frac=0.5)]
    pb = pack_class_matrix(blocks)
    for het_only in (True, False):
        ora = c_oracle.run(pb, min_common=3, het_only=het_only)
        res = engine.run(pb, min_common=3, het_only=het_only, emit_counts=True)
        assert_same_as_oracle(res, ora)
        assert (res.info['n_mfma_tiles'] > 0) == (kernel != 'valu')
        assert res.info['mfma_dtype'] == {'valu': 0, 'mfma': 2, 'mfma_i8': 1}[kernel]


def test_mixed_batch_uses_both_kinds_of_count_kernel(engine):
    """one run whose blocks go to different count kernels (a large dense block on the matrix cores, small and
    banded blocks on the VALU popcount kernel): rows, counts and means must match the oracle across the seams"""
    from oracle import c_oracle
    from util_synth import pack_class_matrix, random_block
    rng = np.random.Generator(np.random.PCG64(4242))
    blocks = [random_block(rng, 30, 150, tri_frac=0.2, het_frac=0.4),
              random_block(rng, 260, 4200, tri_frac=0.1, het_frac=0.5),
              random_block(rng, 120, 900, banded=True, tri_frac=0.1, het_frac=0.3),
              random_block(rng, 3, 40, het_frac=1.0)]
    pb = pack_class_matrix(blocks)
    for het_only in (True, False):
        ora = c_oracle.run(pb, min_common=4, het_only=het_only, n_shuffles=40, seed=17)
        res = engine.run(pb, min_common=4, het_only=het_only, emit_counts=True, n_shuffles=40, seed=17)
        assert_same_as_oracle(res, ora)
        assert_perm_same(res, ora, 40)
        assert res.info['n_mfma_tiles'] > 0 and res.info['n_count_launches'] == 2


def test_auto_kernel_choice_uses_matrix_cores_on_large_dense_blocks(engine):
    import lgmi
    from oracle import c_oracle
    spec = lgmi.default_synth_spec(700, 6000, seed=9)
    spec.tri_per_1024 = 100
    db = engine.synth_dense(spec)
    dr = engine.run_device(db, min_common=6, het_only=True, emit_counts=True)
    assert dr.info()['n_mfma_tiles'] > 0 and dr.info()['mfma_dtype'] == 2      # < 2^24 reads: FP4 operands
    res = dr.fetch()
    ora = c_oracle.run(db.download(), min_common=6, het_only=True)
    assert_same_as_oracle(res, ora)
    dr.free()
    db.free()


@pytest.mark.parametrize('n_reads,dtype', [(2 ** 24 - 64, 2), (2 ** 24 + 64, 1)])
def test_matrix_core_kernels_at_the_exactness_limit(engine, n_reads, dtype):
    """a block just below 2^24 reads runs on the FP4 matrix cores (f32 sums of {0,1} products, exact below 2^24:
    the common-read counts here reach ~13.6 million, far above 2^23 where f32 spacing becomes 1); just above,
    the int8 kernel (exact int32 sums) takes over.  Counts must match the CPU oracle bit for bit either way."""
    import lgmi
    from oracle import c_oracle
    spec = lgmi.default_synth_spec(200, n_reads, seed=11)
    spec.tri_per_1024 = 60
    db = engine.synth_dense(spec)
    dr = engine.run_device(db, min_common=6, het_only=False, emit_counts=True)      # 200 x 200: enough rows for a matrix-core tile
    info = dr.info()
    assert info['n_mfma_tiles'] > 0 and info['mfma_dtype'] == dtype
    res = dr.fetch()
    assert res.row_counts.reshape(-1, 9).sum(axis=1).max() > 2 ** 23
    ora = c_oracle.run(db.download(), min_common=6, het_only=False)
    assert_same_as_oracle(res, ora)
    dr.free()
    db.free()


# ---------------------------------------------------------------- API robustness
def test_bad_arguments_are_rejected(engine):
    import ctypes as C
    from lgmi import _lib
    pb = random_batch(5, n_blocks=1)
    st = pb.as_struct()
    lib = engine.lib
    # malformed batch: band beyond the block's reads
    bad = pb.site_n_words.copy()
    bad[0] = 10_000
    st2 = pb.as_struct()
    st2.site_n_words = bad.ctypes.data_as(_lib.u32p)
    h = C.c_void_p()
    assert lib.lgmi_batch_upload(engine.handle, C.byref(st2), C.byref(h)) == _lib.E_ARG
    assert b'band' in lib.lgmi_last_error()
    # positions must increase inside a block
    pos = pb.site_pos.copy()
    if len(pos) > 1:
        pos[1] = pos[0]
        st3 = pb.as_struct()
        st3.site_pos = pos.ctypes.data_as(_lib.i64p)
        assert lib.lgmi_batch_upload(engine.handle, C.byref(st3), C.byref(h)) == _lib.E_ARG
    # reserved bytes / absurd shuffle counts
    prm = lgmi_params(n_shuffles=2**24 + 1)
    res, info = _lib.Result(), _lib.RunInfo()
    assert lib.lgmi_run(engine.handle, C.byref(st), C.byref(prm), C.byref(res), C.byref(info)) == _lib.E_ARG
    assert lib.lgmi_run(None, C.byref(st), C.byref(prm), C.byref(res), C.byref(info)) == _lib.E_ARG


def lgmi_params(**kw):
    import lgmi
    return lgmi.make_params(**kw)


def test_engine_close_releases_resident_objects():
    import lgmi
    eng = lgmi.Engine(0)
    db = eng.upload(random_batch(6, n_blocks=1))
    dr = eng.run_device(db, min_common=2)
    eng.close()                      # frees db and dr first
    assert db.handle is None and dr.handle is None
    db.free()
    dr.free()                        # idempotent, no use-after-free
    with pytest.raises(RuntimeError):
        eng.run(random_batch(6, n_blocks=1))


def test_mi_log_against_correctly_rounded_values(engine):
    """emit.hip's table-driven logarithm (mi_log: the nine logarithms of every emitted row) on the device against the
    correctly rounded value (decimal, 50 digits) for the arguments it can get — integers >= 1 held in doubles: counts
    below 2^24 and products of two counts.  At most one ulp off, and that rarely; ln 1 is exactly 0."""
    from decimal import Decimal, getcontext
    getcontext().prec = 50
    rng = np.random.default_rng(11)
    xs = np.concatenate([
        np.arange(1, 20001),                                              # every small count
        rng.integers(1, 1 << 24, 15000),                                  # counts
        rng.integers(1, 1 << 24, 10000) * rng.integers(1, 1 << 24, 10000),   # margin products, < 2^48
        2 ** np.arange(0, 53), 2 ** np.arange(1, 53) - 1, 2 ** np.arange(1, 52) + 1,
    ]).astype(np.float64)
    got = engine.selftest_log(xs)
    want = np.array([float(Decimal(int(v)).ln()) for v in xs])           # float(Decimal) rounds to nearest
    assert got[0] == 0.0 and np.all(got[xs == 1.0] == 0.0)
    ulps = np.abs(got.view(np.int64) - want.view(np.int64))
    assert ulps.max() <= 1, (ulps.max(), xs[np.argmax(ulps)])
    assert np.mean(ulps == 0) > 0.995, np.mean(ulps == 0)
    # the reference's own logarithm (numpy -> libm) differs from the correctly rounded value about as rarely
    assert np.mean(np.log(xs) == want) > 0.99


def test_row_p_array_is_optional_and_equal_to_the_derived_one(engine):
    """lgmi_params.no_row_p: Monte-Carlo p is (1 + exceed) / (S + 1); the library can leave the array out (what the
    Python host asks for: MIResult.row_p derives it) — with the array, without it and through the two-call form the
    numbers are the same, and exact_2x2 rows keep their array either way"""
    pb = random_batch(77, n_blocks=3, P=(4, 60), R=(20, 500), tri_frac=0.3)
    kw = dict(min_common=3, n_shuffles=40, seed=3, het_only=False)
    with_arr = engine.run(pb, no_row_p=False, **kw)
    without = engine.run(pb, no_row_p=True, **kw)
    assert with_arr._row_p is not None and without._row_p is None and without._p_derived
    np.testing.assert_array_equal(with_arr.row_exceed, without.row_exceed)
    np.testing.assert_array_equal(with_arr.row_p, without.row_p)
    np.testing.assert_array_equal(without.row_p, (1.0 + without.row_exceed) / 41.0)
    db = engine.upload(pb)
    dr = engine.run_device(db, rows_only=True, **kw)
    dr.permute()
    two = dr.fetch()
    np.testing.assert_array_equal(two.row_exceed, without.row_exceed)
    assert two._row_p is None and (two.row_p == without.row_p).all()
    ex = engine.run(pb, exact_2x2=True, no_row_p=True, **kw)
    assert ex._row_p is not None and not ex._p_derived            # exact p is not a function of exceed
    dr.free()
    db.free()


def test_stream_site_base_shifts_the_philox_counters(engine):
    """lgmi_params.stream_site_base: the permutation draws of a pair are keyed by its two site indices PLUS the base, so a
    rank that runs footprints k.. of a run as a batch of its own reproduces the draws of the one batch that holds them
    all — checked against the CPU specification fed the shifted indices"""
    from oracle import c_oracle
    pb = random_batch(4242, n_blocks=3, P=(4, 50), R=(30, 600), tri_frac=0.3)
    kw = dict(min_common=3, n_shuffles=60, seed=11, het_only=True, emit_counts=True)
    db = engine.upload(pb)
    base = 12345
    plain = engine.run_device(db, **kw)
    shifted = engine.run_device(db, stream_site_base=base, **kw)
    a, b = plain.fetch(), shifted.fetch()
    plain.free(); shifted.free(); db.free()
    np.testing.assert_array_equal(a.row_i, b.row_i)
    np.testing.assert_array_equal(a.row_counts, b.row_counts)
    _p, want = c_oracle.perm_rows(b.row_i + np.uint32(base), b.row_j + np.uint32(base), b.row_counts, 60, 11)
    np.testing.assert_array_equal(b.row_exceed, want)
    assert (a.row_exceed != b.row_exceed).any()


@pytest.mark.parametrize('env', [{'LGMI_PERM_ENUM_MAX': '0'}, {'LGMI_PERM_ENUM_MAX': '4096', 'LGMI_PERM_NO_SECOND_LIST': '1'},
                                 {'LGMI_PERM_SIX_PTS': '0', 'LGMI_WORKER_DENSE': '1'}, {'LGMI_PERM_SIX_PTS': '4096', 'LGMI_WORKER_DENSE': '1'},
                                 {'LGMI_PERM_SIX_PTS': '0', 'LGMI_PERM_ENUM_MAX': '0'}],
                         ids=['enumeration_off', 'rows_marked_in_the_queue', 'six_cell_path_off', 'six_cell_gate_wide_open',
                              'every_larger_row_sampled'])
def test_permutation_p_in_the_other_configurations(env):
    """small larger-than-2x2 rows get their tail mass by enumeration (round 3), 3 x 2 / 2 x 3 rows behind a gate by the
    perimeter walk (round 4).  With LGMI_PERM_ENUM_MAX=0 / LGMI_PERM_SIX_PTS=0 — and the same switches in the CPU
    specification — they take the Monte-Carlo paths again (the lock-step loop included, on dense blocks of 40,000 and
    200,000 reads); with LGMI_PERM_NO_SECOND_LIST=1 the enumeration kernel and the six-cell kernel mark their rows in the
    queue instead of listing the others: tests/helpers/perm_enum_worker.py"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'helpers', 'perm_enum_worker.py')],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith('OK') and int(last.split('general=')[1]) > 100


def test_pinned_pool_both_ways_of_pinning_give_the_same_rows():
    """round 4: result buffers from 8 MB on are huge-page memory registered with hipHostRegister (3.7 ms per 400 MB instead
    of 52 - 72 with hipHostMalloc, tools/ubench_pin.hip); LGMI_PINNED=hostmalloc keeps the old way, which is also the
    fallback.  The same runs fetched through either: the same bytes (tests/helpers/pinned_worker.py)"""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    out = []
    for mode in (None, 'hostmalloc'):
        env = {k: v for k, v in os.environ.items() if k != 'LGMI_PINNED'}
        if mode:
            env['LGMI_PINNED'] = mode
        r = subprocess.run([sys.executable, os.path.join(here, 'helpers', 'pinned_worker.py')], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        out.append(r.stdout.strip().splitlines()[-1])
    assert out[0].startswith('OK rows=') and out[0] == out[1], out


def test_host_rows_are_views_that_outlive_the_device_result(engine):
    """round 4: MIResult's arrays are zero-copy views of the library's pinned host buffers (a batch of footprints returns
    hundreds of megabytes); they stay valid after the device result, other results and the MIResult object itself are
    gone, and the buffers go back to the pool with the last view"""
    import gc
    pb = random_batch(77, n_blocks=3)
    dr = engine.run_device(engine.upload(pb), min_common=3, het_only=True, n_shuffles=50, seed=5)
    res = dr.fetch()
    keep_mi, keep_i, keep_e = res.row_mi, res.row_i, res.row_exceed
    copy_mi, copy_i, copy_e = keep_mi.copy(), keep_i.copy(), keep_e.copy()
    assert not keep_mi.flags.owndata and res.n_rows > 10
    dr.free()
    del res
    gc.collect()
    for seed in range(3):                                   # other results come and go through the same pinned pool
        other = engine.run(random_batch(78 + seed, n_blocks=2), min_common=3, het_only=True, n_shuffles=50, seed=5)
        assert other.n_rows >= 0
        del other
    gc.collect()
    np.testing.assert_array_equal(keep_mi, copy_mi)
    np.testing.assert_array_equal(keep_i, copy_i)
    np.testing.assert_array_equal(keep_e, copy_e)
