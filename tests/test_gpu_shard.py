"""GPU side of tile-level sharding: the shards of one batch, run one after the other on the one GPU a
test box has, concatenate to exactly the unsharded result — rows, tables, MI, permutation p — and their per-site
integer sums add up to the unsharded per-site means.  Also the single-rank form of the HBM-to-HBM gather (RCCL
refuses two ranks on one device, so the N > 1 wire path is covered by construction + the CPU gloo tests)."""
import numpy as np
import pytest

from util_synth import pack_class_matrix, random_batch, random_block

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    import lgmi
    eng = lgmi.Engine(0)
    yield eng
    eng.close()


def run_shards(engine, db, world, **kw):
    parts = []
    for r in range(world):
        dr = engine.run_device(db, shard=(r, world), **kw)
        parts.append((dr.fetch(), dr.info()))
        dr.free()
    return parts


def assert_shards_equal_whole(whole, parts, has_p):
    cat = lambda f: np.concatenate([getattr(p, f) for p, _ in parts])
    for f in ('row_i', 'row_j', 'row_mi', 'row_counts') + (('row_p', 'row_exceed') if has_p else ()):
        np.testing.assert_array_equal(cat(f), getattr(whole, f))                 # bit for bit, in order
    np.testing.assert_array_equal(sum(p.site_n_pairs.astype(np.int64) for p, _ in parts), whole.site_n_pairs)
    # a shard's mean covers its own rows; weighted by the counts the shards' means give back the unsharded mean
    num = sum(np.nan_to_num(p.site_mean_mi) * p.site_n_pairs for p, _ in parts)
    m = whole.site_n_pairs > 0
    assert np.max(np.abs(num[m] / whole.site_n_pairs[m] - whole.site_mean_mi[m]), initial=0.0) <= 1e-11
    assert sum(i['n_examined'] for _, i in parts) == parts[0][1]['n_examined_total']
    assert sum(i['n_rows'] for _, i in parts) == whole.n_rows


@pytest.mark.parametrize('world', [2, 3, 8])
def test_shards_of_small_blocks_concatenate_to_the_whole(engine, world):
    pb = random_batch(900 + world, n_blocks=5, P=(2, 90), R=(6, 700), tri_frac=0.3)
    db = engine.upload(pb)
    for het_only in (True, False):
        kw = dict(min_common=3, het_only=het_only, n_shuffles=40, seed=7, emit_counts=True)
        dr = engine.run_device(db, **kw)
        whole = dr.fetch()
        dr.free()
        assert_shards_equal_whole(whole, run_shards(engine, db, world, **kw), True)
    db.free()


def test_shards_of_one_dense_block_on_the_matrix_cores(engine):
    """one block big enough for 128 x 128 matrix-core tiles, cut 4 and 7 ways; each shard computes fewer tiles"""
    import lgmi
    spec = lgmi.default_synth_spec(1500, 9000, seed=31)
    spec.tri_per_1024 = 100
    db = engine.synth_dense(spec)
    kw = dict(min_common=6, het_only=True, n_shuffles=30, seed=5, emit_counts=True)
    dr = engine.run_device(db, **kw)
    whole, winfo = dr.fetch(), dr.info()
    dr.free()
    assert winfo['n_mfma_tiles'] > 0
    for world in (4, 7):
        parts = run_shards(engine, db, world, **kw)
        assert_shards_equal_whole(whole, parts, True)
        assert max(i['n_mfma_tiles'] for _, i in parts) < winfo['n_mfma_tiles']
    db.free()


def test_column_walk_rows_in_several_segments_and_shards(engine):
    """a site that is not an x site has its row — the later x sites, reached by a walk down a column of the slot matrix —
    cut into segments of LGMI_EMIT_SEG_Q = 1024 partners (round 5): 2,600 sites of which every second is a het SNP give
    the early sites two segments; whole run against the C oracle, then 3 and 7 shards (cuts inside a site's segments,
    quads whose members fall into different shards) against the whole, compact gather form included"""
    import lgmi
    from oracle import c_oracle
    spec = lgmi.default_synth_spec(2600, 3000, seed=41)
    spec.het_every = 2
    spec.tri_per_1024 = 60
    db = engine.synth_dense(spec)
    pb = db.download()
    kw = dict(min_common=6, het_only=True, n_shuffles=20, seed=3, emit_counts=True)
    dr = engine.run_device(db, **kw)
    whole = dr.fetch()
    dr.free()
    ora = c_oracle.run(pb, min_common=6, het_only=True, n_shuffles=20, seed=3)
    np.testing.assert_array_equal(whole.row_i, ora['row_i'])
    np.testing.assert_array_equal(whole.row_j, ora['row_j'])
    np.testing.assert_array_equal(whole.row_counts, ora['row_counts'])
    np.testing.assert_array_equal(whole.row_exceed, ora['row_exceed'])
    plan = lgmi.plan_shard(pb, True, (0, 1))
    assert plan['item_seg'][plan['site_xrow'][plan['item_site']] == lgmi._lib.NONE].max() >= 1     # a column-walk row in two segments
    for world in (3, 7):
        parts = run_shards(engine, db, world, **kw)
        assert_shards_equal_whole(whole, parts, True)
    db.free()


def test_shard_arguments_are_checked(engine):
    import lgmi
    pb = random_batch(1, n_blocks=1)
    with pytest.raises(lgmi._lib.LgmiError):
        engine.run(pb, shard=(2, 2))
    with pytest.raises(lgmi._lib.LgmiError):
        engine.run(pb, shard=(1, 0))
    a = engine.run(pb, min_common=1, shard=(0, 1), emit_counts=True)
    b = engine.run(pb, min_common=1, emit_counts=True)
    np.testing.assert_array_equal(a.row_counts, b.row_counts)


def test_single_rank_device_gather_keeps_everything(engine):
    """lgmi_comm_gather with world = 1: the gathered resident result equals the input (rows, p, tables, per-site
    figures), site_base shifts the indices, rank_row_begin brackets the rows"""
    pb = random_batch(556, n_blocks=2)
    if not engine.world or engine.world == 1:
        try:
            engine.comm_init(engine.comm_unique_id(), 0, 1)
        except Exception as e:                                  # another test of this module's engine did it already
            assert 'already' in str(e)
    db = engine.upload(pb)
    dr = engine.run_device(db, min_common=3, het_only=True, n_shuffles=10, seed=3, emit_counts=True)
    res = dr.fetch()
    for same_batch, base in ((True, 0), (False, 0), (False, 1000)):
        g, begins = engine.comm_gather(dr, root=0, site_base=base, same_batch=same_batch)
        got = g.fetch()
        g.free()
        assert begins == [0, res.n_rows]
        np.testing.assert_array_equal(got.row_i, res.row_i + base)
        np.testing.assert_array_equal(got.row_j, res.row_j + base)
        for f in ('row_mi', 'row_p', 'row_exceed', 'row_counts'):
            np.testing.assert_array_equal(getattr(got, f), getattr(res, f))
        assert len(got.site_mean_mi) == base + len(res.site_mean_mi)
        np.testing.assert_array_equal(got.site_n_pairs[base:], res.site_n_pairs)
        np.testing.assert_array_equal(got.site_mean_mi[base:], res.site_mean_mi)        # NaN-aware equality
        assert (got.site_n_pairs[:base] == 0).all() and np.isnan(got.site_mean_mi[:base]).all()
    dr.free()
    db.free()


def test_split_run_and_two_phase_gather(engine):
    """lgmi_run_device_rows + lgmi_dresult_permute == lgmi_run_device, and the gather started between the two halves
    (begin: sizes + (i, j, mi, tables) on the communication stream; finish: exceed, p, per-site figures) returns the
    same result as the one-call gather — the shape a multi-GPU step uses to hide the row transfer under the
    permutation kernels"""
    pb = random_batch(777, n_blocks=3, tri_frac=0.3)
    try:
        engine.comm_init(engine.comm_unique_id(), 0, 1)
    except Exception as e:
        assert 'already' in str(e)
    db = engine.upload(pb)
    kw = dict(min_common=3, het_only=True, n_shuffles=25, seed=9, emit_counts=True)
    whole = engine.run_device(db, **kw)
    ref = whole.fetch()
    whole.free()
    for counts in (True, False):
        kw['emit_counts'] = counts
        dr = engine.run_device(db, rows_only=True, **kw)
        assert dr.info()['ms_perm'] < 0.5 and dr.info()['n_rows'] == ref.n_rows
        flight = engine.comm_gather_begin(dr, root=0, same_batch=True)
        dr.permute()
        assert dr.info()['n_general_rows'] == ref.info['n_general_rows']
        g, begins = flight.finish()
        got, own = g.fetch(), dr.fetch()
        g.free()
        dr.free()
        assert begins == [0, ref.n_rows]
        for res in (got, own):
            for f in ('row_i', 'row_j', 'row_mi', 'row_p', 'row_exceed', 'site_n_pairs', 'site_mean_mi'):
                np.testing.assert_array_equal(getattr(res, f), getattr(ref, f))
            if counts:
                np.testing.assert_array_equal(res.row_counts, ref.row_counts)
            else:
                assert res.row_counts is None
    dr = engine.run_device(db, rows_only=True, min_common=3, het_only=True)       # no p requested: permute is a no-op
    dr.permute()
    np.testing.assert_array_equal(dr.fetch().row_mi, ref.row_mi)
    dr.free()
    db.free()


def test_one_call_splits_itself_under_a_small_memory_budget(engine, monkeypatch):
    """lgmi_run_device / lgmi_run cut a run that does not fit the memory budget into sequential shards on the one GPU
    (LGMI_MEM_BUDGET_MB forces it here): rows, tables, MI, exceed and the per-site figures are bit-equal to the
    single-sequence run, the info says how many shards ran"""
    import lgmi
    spec = lgmi.default_synth_spec(1500, 9000, seed=33)
    spec.tri_per_1024 = 100
    db = engine.synth_dense(spec)
    kw = dict(min_common=6, het_only=True, n_shuffles=30, seed=5)
    fields = ('row_i', 'row_j', 'row_mi', 'row_exceed', 'row_p', 'site_n_pairs', 'site_mean_mi')
    for counts in (False, True):
        monkeypatch.setenv('LGMI_NO_AUTO_SPLIT', '1')
        dr = engine.run_device(db, emit_counts=counts, **kw)
        whole, winfo = dr.fetch(), dr.info()
        dr.free()
        assert winfo['n_seq_shards'] == 1
        monkeypatch.delenv('LGMI_NO_AUTO_SPLIT')
        monkeypatch.setenv('LGMI_MEM_BUDGET_MB', '40')       # slots + operands alone are ~14 MB here
        dr = engine.run_device(db, emit_counts=counts, **kw)
        got, ginfo = dr.fetch(), dr.info()
        dr.free()
        assert ginfo['n_seq_shards'] > 1 and ginfo['n_rows'] == winfo['n_rows'] and ginfo['n_examined'] == winfo['n_examined']
        assert ginfo['n_general_rows'] == winfo['n_general_rows']
        for f in fields + (('row_counts',) if counts else ()):
            np.testing.assert_array_equal(getattr(got, f), getattr(whole, f), err_msg=f)
        # the host-to-host call takes the same way
        pb = db.download()
        one = engine.run(pb, emit_counts=counts, **kw)
        assert one.info['n_seq_shards'] > 1
        for f in fields:
            np.testing.assert_array_equal(getattr(one, f), getattr(whole, f), err_msg='lgmi_run ' + f)
        monkeypatch.delenv('LGMI_MEM_BUDGET_MB')
    db.free()
