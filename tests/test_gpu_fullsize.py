"""Full-size GPU checks (BASELINE.json sizes) through properties that do not need a CPU
recomputation of every pair: a random sample of rows against numpy popcounts on the
downloaded planes, ordering, margins, mean-MI consistency, p-value laws."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def popcount64(a):
    a = a.astype(np.uint64)
    a = a - ((a >> np.uint64(1)) & np.uint64(0x5555555555555555))
    a = (a & np.uint64(0x3333333333333333)) + ((a >> np.uint64(2)) & np.uint64(0x3333333333333333))
    a = (a + (a >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return ((a * np.uint64(0x0101010101010101)) >> np.uint64(56)).astype(np.int64)


def site_class_planes(pb, s):
    off, nw = int(pb.site_plane_off[s]), int(pb.site_n_words[s])
    lo, hi = pb.planes[off:off + nw], pb.planes[off + nw:off + 2 * nw]
    return [lo & hi, lo & ~hi, hi & ~lo]          # class 0, 1, 2


def mi_numpy(t):
    t = t.astype(float)
    n = t.sum()
    r, c = t.sum(1), t.sum(0)
    if (r > 0).sum() <= 1 or (c > 0).sum() <= 1:
        return 0.0
    m = 0.0
    for a in range(3):
        for b in range(3):
            if t[a, b]:
                m += t[a, b] / n * np.log(n * t[a, b] / (r[a] * c[b]))
    return max(m, 0.0)


@pytest.fixture(scope='module')
def engine():
    import lgmi
    eng = lgmi.Engine(0)
    yield eng
    eng.close()


@pytest.mark.parametrize('n_sites,n_reads,n_shuffles', [(10_000, 50_000, 1000), (50_000, 200_000, 0)],
                         ids=['cfg2_10kx50k_S1000', 'north_star_50kx200k'])
def test_full_size_properties(engine, n_sites, n_reads, n_shuffles):
    import lgmi
    spec = lgmi.default_synth_spec(n_sites, n_reads, seed=20250808)
    db = engine.synth_dense(spec)
    dr = engine.run_device(db, min_common=6, het_only=True, n_shuffles=n_shuffles, seed=11, emit_counts=True)
    info = dr.info()
    res = dr.fetch()
    dr.free()
    pb = db.download()
    db.free()
    t = pb.site_type
    het = t == 2
    H, P = int(het.sum()), n_sites
    # every het-involved pair of a dense block has ~0.81 R common reads: all are emitted, once, in order
    assert info['n_examined'] == H * (P - H) + H * (H - 1) // 2 == res.n_rows
    i, j = res.row_i.astype(np.int64), res.row_j.astype(np.int64)
    assert (i < j).all() and (het[i] | het[j]).all()
    key = i * P + j
    assert (np.diff(key) > 0).all()                                  # sorted by (i, j), no duplicates
    # margins: table total = common reads, bounded by both depths; class-0 cells only at tri sites
    n = res.row_counts.sum(axis=(1, 2))
    assert n.min() >= 6 and 0.78 * n_reads < n.mean() < 0.84 * n_reads
    rng = np.random.default_rng(3)
    sample = rng.choice(res.n_rows, 1500, replace=False)
    planes = {}
    for r in sample:
        a, b = int(i[r]), int(j[r])
        for s in (a, b):
            if s not in planes:
                planes[s] = site_class_planes(pb, s)
        tab = np.array([[popcount64(planes[a][x] & planes[b][y]).sum() for y in range(3)] for x in range(3)])
        assert (tab == res.row_counts[r]).all(), (a, b)               # counts: bit-exact
        assert abs(mi_numpy(tab) - res.row_mi[r]) <= 1e-6
    # MI is symmetric in the two sites and bounded by ln 3
    assert res.row_mi.min() >= 0.0 and res.row_mi.max() <= np.log(3) + 1e-12
    # per-site mean MI equals the mean over the rows touching the site
    sums = np.bincount(i, res.row_mi, P) + np.bincount(j, res.row_mi, P)
    cnts = np.bincount(i, minlength=P) + np.bincount(j, minlength=P)
    np.testing.assert_array_equal(cnts, res.site_n_pairs)
    m = cnts > 0
    assert np.max(np.abs(sums[m] / cnts[m] - res.site_mean_mi[m])) <= 1e-9
    # linked het pairs carry far more information than independent pairs
    hh = het[i] & het[j]
    assert res.row_mi[hh].mean() > 20 * res.row_mi[~hh].mean()
    if n_shuffles:
        S = n_shuffles
        np.testing.assert_array_equal(res.row_p, (1.0 + res.row_exceed) / (S + 1.0))
        assert (res.row_exceed[hh] == 0).all()                        # haplotype-linked: never matched by a shuffle
        # independent pairs: p-values are (discretely) uniform — Kolmogorov distance on a sample of 2e5 rows
        pn = res.row_p[~hh]
        pn = pn[rng.choice(len(pn), 200_000, replace=False)]
        d = np.max(np.abs(np.sort(pn) - (np.arange(len(pn)) + 0.5) / len(pn)))
        assert d < 0.01, d
        assert 0.45 < pn.mean() < 0.55
