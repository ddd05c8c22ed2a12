set -u
timeout -k 10 900 python -m pytest tests/test_gpu_shard.py tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_gather2.py tests/test_cli.py "tests/test_gpu_fullsize.py::test_full_size_properties" "tests/test_gpu_fullsize.py::test_footprint_shaped_batch_against_the_c_oracle" -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ops.json 2> gpurun_out/ops.err
python -c "import json; d=json.load(open('gpurun_out/ops.json')); print({k: round(v,2) for k,v in d['stage_ms'].items() if k in ('ms_total','ms_prep','ms_count','ms_emit','ms_perm')}, round(d['ms_per_step'],2), d['value'], round(d['host_to_host']['ms'],1))"
timeout -k 10 400 python tools/shard_times.py 1 2 4 8 2>&1 | tail -4
