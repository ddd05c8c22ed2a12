cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-host-to-host --steps 8 --warmup 2 > gpurun_out/conc_single.json 2> gpurun_out/conc_single.err
python bench.py --no-cpu-baseline --no-host-to-host --steps 12 --warmup 3 > gpurun_out/conc_a.json 2> gpurun_out/conc_a.err &
PA=$!
python bench.py --no-cpu-baseline --no-host-to-host --steps 12 --warmup 3 > gpurun_out/conc_b.json 2> gpurun_out/conc_b.err &
PB=$!
wait $PA; wait $PB
python - <<'P'
import json
for n in ('single','a','b'):
    d=json.load(open('gpurun_out/conc_%s.json'%n)); print(n, round(d['ms_per_step'],1), [round(x) for x in d['step_wall_ms']], {k:round(v,1) for k,v in d['stage_ms'].items() if k in ('ms_count','ms_perm','ms_emit','ms_total')})
P
