"""One rank of tests/test_gpu_gather2.py: N processes share GPU 0; RCCL is replaced by tests/helpers/fake_rccl.cpp
(LGMI_RCCL_LIB), everything else — liblgmi's gather code, the engine, the socket rendezvous — is the product's.
Rank 0 compares every gathered array with what one unsharded run gives and prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'l-giremi_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import lgmi                                    # noqa: E402
from lgmi.dist import group_from_env           # noqa: E402
from util_synth import random_batch            # noqa: E402

ROW_FIELDS = ('row_i', 'row_j', 'row_mi', 'row_p', 'row_exceed')


def same(got, ref, counts, what):
    for f in ROW_FIELDS + (('row_counts',) if counts else ()) + ('site_n_pairs', 'site_mean_mi'):
        np.testing.assert_array_equal(getattr(got, f), getattr(ref, f), err_msg='%s: %s' % (what, f))   # NaN-aware, bit for bit


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    group = group_from_env()
    eng = lgmi.Engine(0)
    eng.comm_init_group(group)
    assert eng.world == world
    ci = eng.comm_info()      # the communicator's own account of itself; the stand-in is reported as one
    assert ci['stand_in'] and ci['initialised'] and ci['nranks'] == world and ci['rank'] == rank, ci
    assert ci['world_given'] == world and ci['rank_given'] == rank and 'fake' in ci['lib_path'], ci
    checked = []

    # ---- (1) strong scaling: every rank holds the SAME dense block (tri-allelic sites, matrix-core tiles), runs its
    #          shard; rows, tables, exceed counts travel to rank 0, per-site integer sums are reduced
    spec = lgmi.default_synth_spec(1500, 9000, seed=31)
    spec.tri_per_1024 = 100
    db = eng.synth_dense(spec)
    for counts in (True, False):
        kw = dict(min_common=6, het_only=True, n_shuffles=30, seed=5, emit_counts=counts)
        whole = eng.run_device(db, **kw)
        ref = whole.fetch()
        whole.free()
        dr = eng.run_device(db, shard=(rank, world), **kw)
        mine = dr.info()['n_rows']
        g, begins = eng.comm_gather(dr, root=0, site_base=0, same_batch=True)
        assert len(begins) == world + 1 and begins[0] == 0 and begins[-1] == ref.n_rows
        assert begins[rank + 1] - begins[rank] == mine
        if rank == 0:
            same(g.fetch(), ref, counts, 'one-call gather, counts=%s' % counts)
            g.free()
        else:
            assert g is None
        dr.free()
        # the same with the transfer started between the rows and the permutation stage
        dr = eng.run_device(db, shard=(rank, world), rows_only=True, **kw)
        flight = eng.comm_gather_begin(dr, root=0, same_batch=True)
        dr.permute()
        g, begins2 = flight.finish()
        assert begins2 == begins
        if rank == 0:
            same(g.fetch(), ref, counts, 'two-phase gather, counts=%s' % counts)
            g.free()
        dr.free()
        checked.append('same_batch counts=%s rows=%d' % (counts, ref.n_rows))
    db.free()

    # ---- (1b) the same on banded blocks where many pairs fall below min_common: a rank that skipped a pair sends its
    #           partners (row_j), one that emitted all of its pairs does not (compact gather: comm.cpp) — several blocks,
    #           so that some shards are of either kind; the gathered result also has a compact form of its own
    pbb = random_batch(5200, n_blocks=5, P=(30, 260), R=(200, 2500), banded=True, tri_frac=0.2)
    dense_small = random_batch(5201, n_blocks=2, P=(120, 200), R=(300, 600), banded=False)
    from lgmi.pack import concat_batches
    dbb = eng.upload(concat_batches([dense_small, pbb, dense_small]))
    for het_only in (True, False):
        kw = dict(min_common=6, het_only=het_only, n_shuffles=30, seed=5, emit_counts=False)
        whole = eng.run_device(dbb, **kw)
        ref = whole.fetch()
        whole.free()
        dr = eng.run_device(dbb, shard=(rank, world), **kw)
        g, begins = eng.comm_gather(dr, root=0, site_base=0, same_batch=True)
        if rank == 0:
            same(g.fetch(), ref, False, 'banded one-call gather, het_only=%s' % het_only)
            if os.environ.get('LGMI_TEST_GATHER_FORM') == 'plain':
                try:                                           # the plain gather carries no per-site row counts: no compact form
                    g.fetch(compact=True)
                    raise AssertionError('a compact fetch of a plainly gathered result was accepted')
                except lgmi._lib.LgmiError as e:
                    assert e.code == lgmi._lib.E_STATE, str(e)
            else:
                c = g.fetch(compact=True)
                assert c.compact
                same(c, ref, False, 'banded gather, compact fetch of the gathered result')
            g.free()
        dr.free()
        checked.append('banded same_batch het_only=%s rows=%d' % (het_only, ref.n_rows))
    dbb.free()

    # ---- (2) blocks dealt to ranks: every rank runs ITS OWN batch; site indices are shifted by the rank's site base,
    #          per-site figures are concatenated
    pbs = [random_batch(4100 + r, n_blocks=3, P=(2, 80), R=(6, 600), tri_frac=0.3) for r in range(world)]
    bases = np.concatenate([[0], np.cumsum([len(pb.site_pos) for pb in pbs])]).astype(int)
    kw = dict(min_common=3, het_only=True, n_shuffles=25, seed=9, emit_counts=True)
    my = eng.upload(pbs[rank])
    dr = eng.run_device(my, **kw)
    g, begins = eng.comm_gather(dr, root=0, site_base=int(bases[rank]), same_batch=False)
    if rank == 0:
        got = g.fetch()
        g.free()
        for r in range(world):                                 # what rank r computed, recomputed here
            d = eng.upload(pbs[r])
            x = eng.run_device(d, **kw)
            ref = x.fetch()
            x.free()
            d.free()
            lo, hi = begins[r], begins[r + 1]
            assert hi - lo == ref.n_rows
            np.testing.assert_array_equal(got.row_i[lo:hi], ref.row_i + bases[r])
            np.testing.assert_array_equal(got.row_j[lo:hi], ref.row_j + bases[r])
            for f in ('row_mi', 'row_p', 'row_exceed', 'row_counts'):
                np.testing.assert_array_equal(getattr(got, f)[lo:hi], getattr(ref, f), err_msg='rank %d %s' % (r, f))
            np.testing.assert_array_equal(got.site_n_pairs[bases[r]:bases[r + 1]], ref.site_n_pairs)
            np.testing.assert_array_equal(got.site_mean_mi[bases[r]:bases[r + 1]], ref.site_mean_mi)
        assert len(got.site_mean_mi) == bases[-1]
        checked.append('own batches rows=%d sites=%d' % (begins[-1], bases[-1]))
    dr.free()
    my.free()

    # ---- (3) a rank with nothing to send (an empty shard) does not stall the others
    tiny = eng.upload(random_batch(7, n_blocks=1, P=(3, 4), R=(20, 30)))
    kw = dict(min_common=1, het_only=False, n_shuffles=5, seed=1, emit_counts=True)
    whole = eng.run_device(tiny, **kw)
    ref = whole.fetch()
    whole.free()
    dr = eng.run_device(tiny, shard=(rank, world), **kw)
    g, begins = eng.comm_gather(dr, root=0, site_base=0, same_batch=True)
    if rank == 0:
        same(g.fetch(), ref, True, 'tiny batch')
        g.free()
        checked.append('tiny rows=%d per rank %s' % (ref.n_rows, [begins[k + 1] - begins[k] for k in range(world)]))
    dr.free()
    tiny.free()

    # ---- (4) ranks that disagree (tables on one rank only) all get the error, before anything is posted: nobody hangs,
    #          and the communicator still works afterwards
    d = eng.upload(random_batch(8, n_blocks=2))
    dr = eng.run_device(d, min_common=2, het_only=True, n_shuffles=5, seed=1, emit_counts=(rank == 0))
    try:
        eng.comm_gather(dr, root=0, site_base=0, same_batch=True)
        raise AssertionError('mismatched flags were accepted')
    except lgmi._lib.LgmiError as e:
        assert 'ranks disagree' in str(e), str(e)
    dr.free()
    dr = eng.run_device(d, min_common=2, het_only=True, n_shuffles=5, seed=1, emit_counts=True, shard=(rank, world))
    g, begins = eng.comm_gather(dr, root=0, site_base=0, same_batch=True)
    if rank == 0:
        x = eng.run_device(d, min_common=2, het_only=True, n_shuffles=5, seed=1, emit_counts=True)
        same(g.fetch(), x.fetch(), True, 'after a refused gather')
        x.free()
        g.free()
        checked.append('mismatch refused on every rank')
    dr.free()
    d.free()

    group.barrier()
    eng.close()
    if rank == 0:
        print(json.dumps({'ok': True, 'world': world, 'checked': checked, 'comm_info': ci}))


if __name__ == '__main__':
    main()
