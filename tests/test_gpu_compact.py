"""The compact row form (include/lgmi.h, ABI 6) against the plain one: row_begin / site_row_full / row_j_listed /
row_exceed16 must expand to exactly the arrays the plain fetch ships — on the reference-generated golden cases, on a
dense block (every candidate emitted: no partner is listed), on banded blocks (pairs skipped by min_common: partners
listed), with all pairs (het_only = 0), through lgmi_run's pipelined copy with the permutation stage cut into row
ranges, with exact p, and through a run that splits itself into sequential shards.  Row contract:
src/giremi/mutual_information.py:42-45 (rows [p1, type1, p2, type2, mi] in combinations order)."""
import os

import numpy as np
import pytest

from conftest import all_pair_cases, sites_to_mismatches
from util_synth import random_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def engine():
    import lgmi
    eng = lgmi.Engine(0)
    yield eng
    eng.close()


def assert_same(a, b, has_p, exact=False):
    """a: compact, b: plain"""
    assert a.compact and not b.compact
    assert a.n_rows == b.n_rows
    rb = a.row_begin
    assert len(rb) == len(a.site_mean_mi) + 1 and rb[0] == 0 and rb[-1] == a.n_rows and (np.diff(rb.astype(np.int64)) >= 0).all()
    listed = int(np.diff(rb.astype(np.int64))[a.site_row_full == 0].sum())
    assert listed == len(a.row_j_listed)
    for f in ('row_i', 'row_j', 'row_mi', 'site_n_pairs'):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f), err_msg=f)
    np.testing.assert_array_equal(a.site_mean_mi, b.site_mean_mi)            # (NaN == NaN here)
    if b.row_counts is not None:
        np.testing.assert_array_equal(a.row_counts, b.row_counts)
    if has_p:
        np.testing.assert_array_equal(a.row_exceed, b.row_exceed)
        np.testing.assert_array_equal(a.row_p, b.row_p)
        if not exact and a.n_shuffles <= 65535:
            assert a._compact['row_exceed16'] is not None and a._compact['row_exceed16'].dtype == np.uint16


def test_golden_cases_expand_to_the_plain_rows(engine):
    from lgmi.pack import pack_blocks
    n = 0
    for case in all_pair_cases():
        if len(case['sites']) < 2 or any(len(s[2]) < 2 for s in case['sites']):
            continue
        pb = pack_blocks([sites_to_mismatches(case['sites'])])
        for kw in (dict(n_shuffles=0), dict(n_shuffles=40, seed=3)):
            mc = max(case['min_common'], 1)
            a = engine.run(pb, min_common=mc, het_only=False, compact=True, **kw)
            b = engine.run(pb, min_common=mc, het_only=False, compact=False, **kw)
            assert_same(a, b, bool(kw['n_shuffles']))
            n += 1
    assert n >= 80


def test_dense_block_lists_no_partner(engine):
    import lgmi
    db = engine.synth_dense(lgmi.default_synth_spec(3000, 20000, seed=11))
    for het_only in (True, False):
        dr = engine.run_device(db, min_common=6, het_only=het_only, n_shuffles=100, seed=9)
        a, b = dr.fetch(compact=True), dr.fetch()
        dr.free()
        assert_same(a, b, True)
        assert len(a.row_j_listed) == 0 and a.site_row_full.all()           # every candidate pair has >= 6 common reads
        assert a.n_rows == dr_rows(3000, het_only)
    db.free()


def dr_rows(P, het_only):
    H = (P + 4) // 5                                                          # every 5th site is a het SNP
    return P * (P - 1) // 2 if not het_only else H * (P - H) + H * (H - 1) // 2


@pytest.mark.parametrize('seed', [1, 2])
def test_banded_blocks_list_the_partners_of_sites_with_skipped_pairs(engine, seed):
    pb = random_batch(4400 + seed, n_blocks=6, P=(40, 300), R=(200, 3000), banded=True, tri_frac=0.2)
    for het_only in (True, False):
        kw = dict(min_common=6, het_only=het_only, n_shuffles=30, seed=seed, emit_counts=True)
        a = engine.run(pb, compact=True, **kw)
        b = engine.run(pb, compact=False, **kw)
        assert_same(a, b, True)
        assert 0 < len(a.row_j_listed) <= a.n_rows and not a.site_row_full.all()
        db = engine.upload(pb)
        dr = engine.run_device(db, **kw)
        assert_same(dr.fetch(compact=True), dr.fetch(), True)
        dr.free()
        db.free()


def test_mixed_full_and_listed_sites_in_one_block(engine):
    """dense coverage except a few shallow sites: most sites are full, the shallow ones' pairs fall below min_common"""
    from util_synth import pack_class_matrix, random_block
    rng = np.random.Generator(np.random.PCG64(77))
    pos, typ, cls = random_block(rng, 220, 900, banded=False, cover=0.9)
    for s in (10, 57, 58, 140, 219):
        cls[s, 8:] = -1                                                       # 8 reads left: most partners share < 6 of them
    pb = pack_class_matrix([(pos, typ, cls)])
    for het_only in (True, False):
        a = engine.run(pb, min_common=6, het_only=het_only, n_shuffles=20, seed=1, compact=True)
        b = engine.run(pb, min_common=6, het_only=het_only, n_shuffles=20, seed=1, compact=False)
        assert_same(a, b, True)
        assert a.site_row_full.any() and not a.site_row_full.all()


def test_exact_p_keeps_32_bit_counts_and_the_p_array(engine):
    pb = random_batch(91, n_blocks=3, P=(30, 80), R=(100, 600), tri_frac=0.3)
    for ns in (0, 25):
        a = engine.run(pb, min_common=4, het_only=True, n_shuffles=ns, seed=4, exact_2x2=True, compact=True)
        b = engine.run(pb, min_common=4, het_only=True, n_shuffles=ns, seed=4, exact_2x2=True, compact=False)
        assert_same(a, b, True, exact=True)
        assert a._compact['row_exceed16'] is None


def test_more_than_65535_shuffles_keep_32_bit_counts(engine):
    pb = random_batch(92, n_blocks=1, P=(12, 12), R=(300, 300), tri_frac=0.0)
    a = engine.run(pb, min_common=4, het_only=False, n_shuffles=70000, seed=4, compact=True)
    b = engine.run(pb, min_common=4, het_only=False, n_shuffles=70000, seed=4, compact=False)
    assert_same(a, b, True, exact=True)
    assert a._compact['row_exceed16'] is None and a.row_exceed.max() > 65535


@pytest.mark.parametrize('chunks', ['1', '3', '7'])
def test_pipelined_copy_with_the_permutation_stage_in_row_ranges(engine, chunks):
    """lgmi_run ships every row range's counts while the next range is computed (LGMI_PERM_CHUNKS): the counts do not
    depend on the cut — Philox counters are keyed by the pair"""
    import lgmi
    db = engine.synth_dense(lgmi.default_synth_spec(2400, 12000, seed=5))      # 1.15e6 rows: >= 65536 per range at 7
    pb = db.download()
    kw = dict(min_common=6, het_only=True, n_shuffles=60, seed=8)
    dr = engine.run_device(db, **kw)
    ref = dr.fetch()
    dr.free()
    db.free()
    old = os.environ.get('LGMI_PERM_CHUNKS')
    os.environ['LGMI_PERM_CHUNKS'] = chunks
    try:
        a = engine.run(pb, compact=True, **kw)
        b = engine.run(pb, compact=False, **kw)
    finally:
        if old is None:
            os.environ.pop('LGMI_PERM_CHUNKS')
        else:
            os.environ['LGMI_PERM_CHUNKS'] = old
    assert_same(a, b, True)
    for f in ('row_i', 'row_j', 'row_mi', 'row_exceed'):
        np.testing.assert_array_equal(getattr(a, f), getattr(ref, f), err_msg=f)
    assert a.info['n_general_rows'] == ref.info['n_general_rows'] and a.info['n_six_rows'] == ref.info['n_six_rows']


def test_a_run_that_splits_itself_has_a_compact_form(engine):
    import lgmi
    db = engine.synth_dense(lgmi.default_synth_spec(2000, 9000, seed=6))
    pb = db.download()
    kw = dict(min_common=6, het_only=True, n_shuffles=20, seed=2)
    dr = engine.run_device(db, **kw)
    ref = dr.fetch()
    dr.free()
    old = os.environ.get('LGMI_MEM_BUDGET_MB')
    os.environ['LGMI_MEM_BUDGET_MB'] = '40'
    try:
        dr = engine.run_device(db, **kw)
        assert dr.info()['n_seq_shards'] > 1
        a, b = dr.fetch(compact=True), dr.fetch()
        dr.free()
        c = engine.run(pb, compact=True, **kw)
    finally:
        if old is None:
            os.environ.pop('LGMI_MEM_BUDGET_MB')
        else:
            os.environ['LGMI_MEM_BUDGET_MB'] = old
    db.free()
    assert_same(a, b, True)
    for r in (a, c):
        for f in ('row_i', 'row_j', 'row_mi', 'row_exceed', 'site_n_pairs'):
            np.testing.assert_array_equal(getattr(r, f), getattr(ref, f), err_msg=f)


def test_views_are_read_only_and_small_arrays_are_copies(engine):
    pb = random_batch(5, n_blocks=2)
    r = engine.run(pb, min_common=3)
    with pytest.raises(ValueError):
        r.row_mi[:1] = 0.0
    r.site_mean_mi[:1] = 0.0                                                  # a copy: the caller's own


def test_empty_results(engine):
    from util_synth import pack_class_matrix
    cls = np.full((3, 10), -1, np.int8)
    pb = pack_class_matrix([(np.array([1, 2, 3]), np.array([2, 0, 0], np.uint8), cls)])
    a = engine.run(pb, min_common=1, n_shuffles=10, compact=True)
    assert a.n_rows == 0 and len(a.row_i) == 0 and len(a.row_j) == 0 and len(a.row_exceed) == 0
    assert (a.row_begin == 0).all()


# ---------------------------------------------------------------- the pipelined upload of lgmi_run (lgmi_batch.site_tri)
def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('pieces', ['1', '3', '8', '64'])
def test_planes_uploaded_in_pieces_under_the_count_kernels(engine, pieces):
    """with the packer's tri flags (lgmi_batch.site_tri) lgmi_run plans before the planes have moved and uploads them in
    pieces of sites, the count kernels of the tiles a piece completes running meanwhile: same rows, tables and counts as
    the plain upload — on small / banded / multi-block batches (VALU tiles), on a dense block (matrix-core tiles, FP4
    operand groups) and with the int8 matrix-core kernel"""
    import lgmi
    from lgmi.pack import concat_batches
    db = engine.synth_dense(lgmi.default_synth_spec(1800, 9000, seed=3))
    dense = db.download()
    db.free()
    assert dense.site_tri is not None and dense.site_tri.any() and not dense.site_tri.all()
    # (batches from random_batch carry no flags: make them the way a packer would)
    def with_flags(pb):
        if pb.site_tri is None:
            tri = np.zeros(pb.n_sites, np.uint8)
            for s in range(pb.n_sites):
                o, w = int(pb.site_plane_off[s]), int(pb.site_n_words[s])
                tri[s] = bool((pb.planes[o:o + w] & pb.planes[o + w:o + 2 * w]).any())
            pb.site_tri = tri
        return pb
    parts = [random_batch(70, n_blocks=7, P=(20, 200), R=(100, 2000), tri_frac=0.3), dense,
             random_batch(71, n_blocks=4, P=(2, 60), R=(6, 500), banded=True)]
    mixed = concat_batches([with_flags(p) for p in parts])
    assert mixed.site_tri is not None
    for pb, extra in ((mixed, {}), (dense, {}), (dense, {'LGMI_COUNT_KERNEL': 'mfma_i8'}), (dense, {'LGMI_COUNT_KERNEL': 'valu'})):
        kw = dict(min_common=5, het_only=True, n_shuffles=25, seed=6, emit_counts=True)
        plain = _with_env(dict(extra, LGMI_NO_UPLOAD_PIPE='1'), lambda: engine.run(pb, **kw))
        piped = _with_env(dict(extra, LGMI_UPLOAD_PIPE_MIN_WORDS='0', LGMI_UPLOAD_CHUNKS=pieces), lambda: engine.run(pb, **kw))
        for f in ('row_i', 'row_j', 'row_mi', 'row_counts', 'row_exceed', 'site_n_pairs', 'site_mean_mi'):
            np.testing.assert_array_equal(getattr(piped, f), getattr(plain, f), err_msg=f)
        assert piped.n_rows > 1000


def test_wrong_tri_flags_are_an_error_not_a_wrong_result(engine):
    import lgmi
    db = engine.synth_dense(lgmi.default_synth_spec(600, 5000, seed=4))
    pb = db.download()
    db.free()
    good = pb.site_tri.copy()
    for flip in (int(np.nonzero(good)[0][0]), int(np.nonzero(good == 0)[0][5])):
        pb.site_tri = good.copy()
        pb.site_tri[flip] ^= 1
        for env in ({'LGMI_UPLOAD_PIPE_MIN_WORDS': '0'}, {'LGMI_NO_UPLOAD_PIPE': '1'}):
            with pytest.raises(lgmi._lib.LgmiError) as e:
                _with_env(env, lambda: engine.run(pb, min_common=5))
            assert e.value.code == lgmi._lib.E_ARG and 'site_tri' in str(e.value)
    pb.site_tri = good
    assert _with_env({'LGMI_UPLOAD_PIPE_MIN_WORDS': '0'}, lambda: engine.run(pb, min_common=5)).n_rows > 0
    # planes that are not packed in site order: the plain upload is taken, the result is the same
    pb2 = lgmi.pack.PackedBatch(pb.block_site_begin, pb.block_n_reads, pb.site_pos, pb.site_type, pb.site_word_off, pb.site_n_words,
                                (pb.site_plane_off + np.uint64(3)), np.concatenate([np.zeros(3, np.uint64), pb.planes]), pb.type_names,
                                pb.bad_sites, good)
    a = _with_env({'LGMI_UPLOAD_PIPE_MIN_WORDS': '0'}, lambda: engine.run(pb2, min_common=5))
    b = engine.run(pb, min_common=5)
    np.testing.assert_array_equal(a.row_mi, b.row_mi)
