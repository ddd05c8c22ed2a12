"""Fixed cost per 128 x 128 tile of k_count_mfma_fp4: the same 50,000-site chromosome (27,886 tiles, 109 per CU) at several
read counts — time = tiles_per_cu x (a + b x words).  `a` is everything a persistent-workgroup variant could hide (workgroup
launch, ring prologue, 256-store epilogue); VERDICT r3 item 4.   python tools/count_tile_overhead.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'l-giremi_amd')]
import lgmi  # noqa: E402


def clocks():
    """the active sclk / mclk levels from sysfs (no child process: see bench.py ClockSampler)"""
    import glob
    out = []
    for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_[sm]clk'))[:2]:
        try:
            out += ['%s %s' % (os.path.basename(f), ln.strip()) for ln in open(f) if '*' in ln]
        except OSError:
            pass
    return out


eng = lgmi.Engine(0)
rows = []
for n_reads in (25_600, 51_200, 102_400, 153_600, 200_000):
    db = eng.synth_dense(lgmi.default_synth_spec(50_000, n_reads, seed=20250808))
    best = None
    for _ in range(3):
        dr = eng.run_device(db, min_common=6, n_shuffles=0, seed=1, het_only=True)
        info = dr.info()
        dr.free()
        best = info if best is None or info['ms_count'] < best['ms_count'] else best
    db.free()
    rows.append({'n_reads': n_reads, 'words': (n_reads + 63) // 64, 'ms_count': best['ms_count'], 'tiles': best['n_mfma_tiles']})
    print(rows[-1], flush=True)
W = np.array([r['words'] for r in rows], float)
t = np.array([r['ms_count'] for r in rows], float)
b, a = np.polyfit(W, t, 1)
tiles = rows[-1]['tiles']
per_cu = tiles / 256.0
out = {'rows': rows, 'fit_ms': {'intercept': a, 'per_word': b}, 'tiles': tiles, 'tiles_per_cu': per_cu,
       'fixed_us_per_tile': 1e3 * a / per_cu, 'us_per_tile_at_north_star': 1e3 * rows[-1]['ms_count'] / per_cu,
       'fixed_share_at_north_star': a / rows[-1]['ms_count'], 'clocks': clocks(),
       'reading': 'ms_count = intercept + per_word x words over the same tile set; intercept / tiles_per_cu = what one tile costs '
                  'beside its MFMAs (launch, prologue, epilogue): the most a persistent-workgroup variant could hide'}
print(json.dumps(out['fit_ms']), out['fixed_us_per_tile'], out['fixed_share_at_north_star'], out['clocks'])
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'count_tile_overhead.json'), 'w'), indent=1)
eng.close()
