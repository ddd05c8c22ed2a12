"""Result buffers of tens of megabytes through the pinned host pool, in a process of its own (the library reads LGMI_PINNED
once): the default path (huge-page memory registered with hipHostRegister) or LGMI_PINNED=hostmalloc (hipHostMalloc, the
path of rounds 1 - 3 and the fallback).  Prints a digest of everything fetched; tests/test_gpu_parity.py compares the two."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]
import lgmi                                        # noqa: E402

eng = lgmi.Engine(0)
h = hashlib.sha256()
rows = 0
db = eng.synth_dense(lgmi.default_synth_spec(6000, 4000, seed=11))
keep = None
for rep in range(3):                               # the second and third fetch reuse the pool's buffers
    res = eng.run_device(db, min_common=6, het_only=True, n_shuffles=100, seed=3 + rep, no_row_p=False).fetch()
    assert res.row_mi.nbytes >= 16 << 20           # beyond the 8 MB from which the pool pins by registration
    for a in (res.row_i, res.row_j, res.row_mi, res.row_exceed, res.row_p):
        h.update(np.ascontiguousarray(a).tobytes())
    rows += res.n_rows
    if rep == 0:
        keep = (res.row_mi, res.row_mi.copy())     # a view that outlives its result while the pool hands buffers out again
    del res
np.testing.assert_array_equal(keep[0], keep[1])
db.free()
eng.close()
print('OK rows=%d sha=%s' % (rows, h.hexdigest()[:32]))
