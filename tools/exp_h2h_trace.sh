# host-side milestones of the one-shot call (lgmi_run) on the north-star chromosome: where host -> host time goes
set -u
mkdir -p gpurun_out
LGMI_TRACE_HOST=1 timeout -k 10 500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/h2h_trace.json 2> gpurun_out/h2h_trace.err
echo rc=$?
grep 'lgmi host' gpurun_out/h2h_trace.err | tail -40
python - <<'PY'
import json
d = json.load(open('gpurun_out/h2h_trace.json'))
print(d['ms_per_step'], d['value'], d['host_to_host'])
PY
