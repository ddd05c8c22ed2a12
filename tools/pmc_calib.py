"""Calibration of FETCH_SIZE for k_count's access pattern (MI355X_MICROARCH.md: "calibrate on a known
byte count in your own access pattern"): a chromosome with ONE x-tile row (<= 64 het sites) and a column
set far larger than the 256 MiB Infinity Cache, so k_count must fetch every column plane exactly once.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_calib.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'l-giremi_amd'))
import lgmi  # noqa: E402

eng = lgmi.Engine(0)
spec = lgmi.default_synth_spec(300, 8_000_000, seed=1)
spec.tri_per_1024 = 0
db = eng.synth_dense(spec)
dr = eng.run_device(db, min_common=6, het_only=True)
info = dr.info()
print('bytes_in', info['bytes_in'], 'n_tile_pairs', info['n_tile_pairs'], 'ms_count', info['ms_count'])
