set -u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=10 > gpurun_out/t_gpu_all.log 2>&1; echo rc=$?; grep -v "^  File\|Extension modules" gpurun_out/t_gpu_all.log | tail -30 | cut -c1-300
LGMI_TRACE_HOST=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; echo bench rc=$?
python -c "
import json; d=json.load(open('gpurun_out/bench_r05.json')); h=d['host_to_host']
print('h2h', round(h['ms'],1), 'ms  min', round(h['ms_min'],1), ' kernels', round(h['kernels_ms'],1), ' d2h_bytes %.3g' % h['d2h_bytes'], ' step', round(d['ms_per_step'],1), d.get('value_host_to_host'), d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])"
