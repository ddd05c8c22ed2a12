set -u
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shard.py tests/test_gpu_compact.py -x -q -m gpu 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > gpurun_out/emit2.json 2> gpurun_out/emit2.err
python -c "import json; d=json.load(open('gpurun_out/emit2.json')); print({k: round(v,2) for k,v in d['stage_ms'].items() if k in ('ms_total','ms_count','ms_emit','ms_perm')})"
( cd build_abl/r4 && timeout -k 10 300 python tools/shard_times.py 1 > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/shard_times.json')); print('r4  ', d['1']['shards'][0])" )
timeout -k 10 300 python tools/shard_times.py 8 2>&1 | tail -1
