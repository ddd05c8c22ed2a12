"""Where a one-shot lgmi_run on footprint-shaped data spends its time: upload, run (first / second), fetch, and the whole
call, each timed on the host.   python tools/fp_oneshot.py [n_footprints]      (LGMI_TRACE_HOST=1 adds the library's own marks)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]
from lgmi.synth import footprint_blocks  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pb = footprint_blocks(n, seed=20250810, cache_dir=os.environ.get('LGMI_BENCH_CACHE', '/tmp'))      # (forks: before the GPU)
import lgmi  # noqa: E402

eng = lgmi.Engine(0)
kw = dict(min_common=6, n_shuffles=1000, seed=20250808, het_only=True)
out = {'n_footprints': n, 'n_sites': int(pb.n_sites), 'plane_MB': pb.planes.nbytes / 1e6}


def t(f):
    t0 = time.perf_counter()
    r = f()
    return r, 1e3 * (time.perf_counter() - t0)


for rep in range(3):
    db, ms_up = t(lambda: eng.upload(pb))
    dr, ms_run1 = t(lambda: eng.run_device(db, **kw))
    _, ms_sync = t(eng.synchronize)
    res, ms_fetch = t(dr.fetch)
    info = dr.info()
    dr.free()
    dr2, ms_run2 = t(lambda: eng.run_device(db, **kw))
    dr2.free(); db.free()
    r1, ms_all = t(lambda: eng.run(pb, **kw))
    out['rep%d' % rep] = {'upload': ms_up, 'run_first': ms_run1, 'sync': ms_sync, 'fetch': ms_fetch, 'run_second': ms_run2,
                          'lgmi_run': ms_all, 'kernels': info['ms_total'], 'plan_host': info['ms_plan_host'], 'rows': int(res.n_rows)}
    print(json.dumps(out['rep%d' % rep]), flush=True)
eng.close()
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'fp_oneshot.json'), 'w'), indent=1)
