"""Seeded random batches with random run parameters — shapes, banded or not, share of tri-allelic sites, min_common,
het_only, shuffle count, compact or plain rows, planes uploaded whole or in pieces, a memory budget that makes the call
split itself, one run or shards put together — through the C ABI against the C
oracle: rows, counts and permutation counts bit for bit, MI within 1e-6.  The suite runs a handful of seeds;
    LGMI_FUZZ_SEEDS=400 python -m pytest tests/test_gpu_fuzz.py -q -m gpu
is the sweep whose log is kept under profiles/ (rNN_parity_fuzz.txt)."""
import os

import numpy as np
import pytest

from util_synth import random_batch

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get('LGMI_FUZZ_SEEDS', '8'))
MI_TOL = 1e-6


@pytest.fixture(scope='module')
def engine():
    import lgmi
    eng = lgmi.Engine(0)
    yield eng
    eng.close()


def _case(seed):
    rng = np.random.Generator(np.random.PCG64(77000 + seed))
    big = rng.random() < 0.15                                    # now and then a block the matrix-core kernel takes
    kw = dict(n_blocks=int(rng.integers(1, 7)), tri_frac=float(rng.choice([0.0, 0.05, 0.3])),
              P=(2, 260) if big else (2, 90), R=(6, 3000) if big else (6, 700),
              banded=[None, True, False][int(rng.integers(0, 3))])
    run = dict(min_common=int(rng.choice([1, 2, 5, 6, 20])), het_only=bool(rng.integers(0, 2)),
               n_shuffles=int(rng.choice([0, 0, 1, 13, 100, 500])), seed=int(rng.integers(0, 2 ** 31)))
    env = {}
    if rng.random() < 0.3:
        env['LGMI_UPLOAD_PIPE_MIN_WORDS'] = '0'                   # lgmi_run brings the planes up in pieces under the count kernels
    if rng.random() < 0.2:
        env['LGMI_MEM_BUDGET_MB'] = str(int(rng.choice([1, 2, 8])))   # ... and cuts itself into sequential shards
    return kw, run, int(rng.integers(1, 6)), bool(rng.integers(0, 2)), env


@pytest.mark.parametrize('seed', range(N_SEEDS))
def test_random_runs_match_the_oracle(engine, seed, monkeypatch):
    from oracle import c_oracle
    kw, run, world, compact, env = _case(seed)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pb = random_batch(88000 + seed, **kw)
    if 'LGMI_UPLOAD_PIPE_MIN_WORDS' in env:
        # the packer's tri flags let the run be planned before the planes are up (lgmi_batch.site_tri): lo & hi anywhere
        off, nw = pb.site_plane_off.astype(np.int64), pb.site_n_words.astype(np.int64)
        pb.site_tri = np.array([(pb.planes[o:o + n] & pb.planes[o + n:o + 2 * n]).any() for o, n in zip(off, nw)], np.uint8)
    ora = c_oracle.run(pb, **run)
    want_p = run['n_shuffles'] > 0

    def check(get, n_rows, what):
        assert n_rows == len(ora['row_i']), what
        np.testing.assert_array_equal(get('row_i'), ora['row_i'], err_msg=what)
        np.testing.assert_array_equal(get('row_j'), ora['row_j'], err_msg=what)
        np.testing.assert_array_equal(get('row_counts'), ora['row_counts'], err_msg=what)
        if n_rows:
            mi = get('row_mi')
            assert np.max(np.abs(mi - ora['row_mi'])) <= MI_TOL and ((mi == 0.0) == (ora['row_mi'] == 0.0)).all(), what
        if run['n_shuffles']:
            np.testing.assert_array_equal(get('row_exceed'), ora['row_exceed'], err_msg=what)
        if want_p:
            np.testing.assert_array_equal(get('row_p'), ora['row_p'], err_msg=what)          # (NaN == NaN here)

    res = engine.run(pb, emit_counts=True, compact=compact, **run)
    check(lambda f: getattr(res, f), res.n_rows, 'one call, %s rows, %s' % ('compact' if compact else 'plain', env))
    np.testing.assert_array_equal(res.site_n_pairs, ora['site_n_pairs'])
    m = ora['site_n_pairs'] > 0
    assert np.isnan(res.site_mean_mi[~m]).all()
    if m.any():
        assert np.max(np.abs(res.site_mean_mi[m] - ora['site_mean_mi'][m])) <= MI_TOL
    if world > 1:
        # the same batch cut into `world` shards (what the ranks of a multi-GPU run compute): their rows, one after the other
        db = engine.upload(pb)
        parts = []
        for r in range(world):
            dr = engine.run_device(db, emit_counts=True, shard=(r, world), **run)
            parts.append(dr.fetch(compact=compact))
            dr.free()
        db.free()
        cat = lambda f: np.concatenate([np.asarray(getattr(p, f)) for p in parts])
        check(cat, sum(p.n_rows for p in parts), '%d shards' % world)
        np.testing.assert_array_equal(sum(np.asarray(p.site_n_pairs, np.int64) for p in parts), ora['site_n_pairs'])
