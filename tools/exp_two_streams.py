"""Two contexts (two HIP streams) of ONE process each running the whole north-star step on its own resident chromosome, from
two host threads: does the VALU-bound permutation stage of one run under the MFMA-bound count kernel of the other?
(tools/exp_two_procs.sh asked the same of two processes: 1.00x.)      python tools/exp_two_streams.py"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]
import lgmi  # noqa: E402

KW = dict(min_common=6, n_shuffles=1000, seed=20250808, het_only=True)
STEPS = 6


def worker(eng, db, out, k, offset_s):
    time.sleep(offset_s)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        eng.run_device(db, **KW).free()
    out[k] = (time.perf_counter() - t0) / STEPS


engs = [lgmi.Engine(0), lgmi.Engine(0)]
dbs = [e.synth_dense(lgmi.default_synth_spec(50_000, 200_000, seed=20250808)) for e in engs]
for e, d in zip(engs, dbs):
    e.run_device(d, **KW).free()                      # warm-up
alone = [None]
worker(engs[0], dbs[0], alone, 0, 0.0)
print('alone: %.1f ms per step' % (1e3 * alone[0]), flush=True)
for offset in (0.0, 0.1):                             # started together / half a step apart
    out = [None, None]
    th = [threading.Thread(target=worker, args=(engs[k], dbs[k], out, k, offset * k)) for k in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0 - offset
    print('two streams (offset %.1f s): %.1f / %.1f ms per step each; %d steps in %.3f s = %.1f ms per step in aggregate (%.2fx)'
          % (offset, 1e3 * out[0], 1e3 * out[1], 2 * STEPS, wall, 1e3 * wall / (2 * STEPS), alone[0] / (wall / (2 * STEPS))), flush=True)
for e, d in zip(engs, dbs):
    d.free()
    e.close()
