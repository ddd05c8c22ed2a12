"""Kernel time of every shard of the north-star chromosome cut N ways, one after the other on one GPU: what each rank
of an N-GPU strong-scaling run computes (the gather is not included).   python tools/shard_times.py [N ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]
import lgmi  # noqa: E402

eng = lgmi.Engine(0)
db = eng.synth_dense(lgmi.default_synth_spec(50_000, 200_000, seed=20250808))
out = {}
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    rows = []
    for r in range(world):
        best = None
        for _ in range(2):                                  # second run: pooled allocations
            dr = eng.run_device(db, min_common=6, n_shuffles=1000, seed=20250808, het_only=True, shard=(r, world))
            info = dr.info()
            dr.free()
            best = info
        rows.append({k: round(best[k], 2) for k in ('ms_total', 'ms_count', 'ms_emit', 'ms_perm', 'ms_prep', 'ms_plan_host')}
                    | {'n_examined': best['n_examined'], 'n_mfma_tiles': best['n_mfma_tiles']})
    out[world] = {'max_ms_total': max(x['ms_total'] for x in rows), 'sum_ms_total': round(sum(x['ms_total'] for x in rows), 1),
                  'shards': rows}
    print(world, out[world]['max_ms_total'], out[world]['sum_ms_total'], flush=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'shard_times.json'), 'w'), indent=1)
eng.close()
