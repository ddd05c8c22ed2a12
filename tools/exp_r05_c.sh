set -u
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_gather2.py tests/test_gpu_shard.py "tests/test_gpu_fullsize.py::test_north_star_without_shuffles" tests/test_gpu_parity.py tests/test_cli.py tests/test_integration_stub.py -x -q -m gpu --durations=8 > gpurun_out/t_c.log 2>&1; echo rc=$?; tail -25 gpurun_out/t_c.log
LGMI_TRACE_HOST=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pipe.json 2> gpurun_out/pipe.err; echo bench rc=$?
grep "run:uploaded" gpurun_out/pipe.err | tail -2; grep "lgmi pipe" gpurun_out/pipe.err | tail -8
python -c "
import json; d=json.load(open('gpurun_out/pipe.json')); h=d['host_to_host']
print('h2h', round(h['ms'],1), 'ms  min', round(h['ms_min'],1), ' kernels', round(h['kernels_ms'],1), ' d2h_bytes %.3g' % h['d2h_bytes'], ' step', round(d['ms_per_step'],1), d.get('value_host_to_host'), d['stage_ms'])"
