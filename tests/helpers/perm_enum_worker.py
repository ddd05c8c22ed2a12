"""GPU parity of the permutation stage in the two configurations the default run does not reach, each in a process of
its own (the library reads its variables once), started by tests/test_gpu_parity.py:
  LGMI_PERM_ENUM_MAX=0     the small-table enumeration switched OFF on both sides (lgo_set_enum_max(0) in the CPU
                           specification): every larger-than-2x2 row takes the Monte-Carlo path — the urn draws and the
                           per-lane state machine that small rows otherwise no longer reach at large S;
  LGMI_PERM_ENUM_MAX=4096 LGMI_PERM_NO_SECOND_LIST=1
                           the enumeration on, but k_perm_enum marking its rows in the queue instead of making the second
                           list (what it does when the queue fills more than half of its buffer); k_perm_six then has no
                           third list either and marks its rows the same way;
  LGMI_PERM_SIX_PTS=0      the exact six-cell path (round 4) switched OFF on both sides (lgo_set_six_pts(0)): every 3 x 2 /
                           2 x 3 row takes the lock-step sampling loop of k_perm_general again — the loop that ran 57 % of
                           the north-star step in round 3 and that the default run now only reaches for rows behind the gate;
                           with LGMI_WORKER_DENSE=1 also on dense synthetic blocks of 40,000 and 200,000 reads;
  LGMI_PERM_SIX_PTS=4096   the gate opened wide: rows of thousands of chords take the perimeter walk."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd'), os.path.join(ROOT, 'tests')]
MODE = os.environ.get('LGMI_PERM_ENUM_MAX', '4096')
assert MODE in ('0', '4096')
SIX = os.environ.get('LGMI_PERM_SIX_PTS', '16')

import lgmi                                        # noqa: E402
from oracle import c_oracle                        # noqa: E402
from util_synth import random_batch                # noqa: E402

lib = c_oracle.load()
lib.lgo_set_enum_max.restype = ctypes.c_uint32
lib.lgo_set_enum_max.argtypes = [ctypes.c_uint32]
lib.lgo_set_enum_max(int(MODE))
lib.lgo_set_six_pts.restype = ctypes.c_uint32
lib.lgo_set_six_pts.argtypes = [ctypes.c_uint32]
lib.lgo_set_six_pts(int(SIX))
eng = lgmi.Engine(0)
rows = general = 0
for seed in range(6):
    pb = random_batch(5200 + seed, n_blocks=1 + seed % 3, tri_frac=0.3, R=(6, 900))
    S = [100, 257, 1000][seed % 3]
    ora = c_oracle.run(pb, min_common=[1, 5][seed % 2], het_only=True, n_shuffles=S, seed=7 + seed)
    res = eng.run(pb, min_common=[1, 5][seed % 2], het_only=True, n_shuffles=S, seed=7 + seed, emit_counts=True)
    np.testing.assert_array_equal(res.row_i, ora['row_i'])
    np.testing.assert_array_equal(res.row_counts, ora['row_counts'])
    np.testing.assert_array_equal(res.row_exceed, ora['row_exceed'])
    c = res.row_counts.reshape(-1, 3, 3)
    general += int((((c.sum(axis=2) > 0).sum(axis=1) > 2) | ((c.sum(axis=1) > 0).sum(axis=1) > 2)).sum())
    rows += res.n_rows
if os.environ.get('LGMI_WORKER_DENSE') == '1':
    for n_reads, S in ((40000, 64), (200000, 150)):
        spec = lgmi.default_synth_spec(60 if n_reads == 40000 else 36, n_reads, seed=5)
        spec.tri_per_1024 = 250
        db = eng.synth_dense(spec)
        pb = db.download()
        res = eng.run_device(db, min_common=6, het_only=True, n_shuffles=S, seed=1234).fetch()
        ora = c_oracle.run(pb, min_common=6, het_only=True, n_shuffles=S, seed=1234)
        np.testing.assert_array_equal(res.row_i, ora['row_i'])
        np.testing.assert_array_equal(res.row_exceed, ora['row_exceed'])
        rows += res.n_rows
        db.free()
eng.close()
print('OK rows=%d general=%d' % (rows, general))
