set -u
for cw in 2000 800 200 0; do
  echo "COST_WALK $cw: $(LGMI_COST_WALK=$cw timeout -k 10 300 python tools/shard_times.py 8 2>&1 | tail -1)"
  python - <<PY
import json
d=json.load(open('gpurun_out/shard_times.json'))['8']
print('   ', [s['ms_total'] for s in d['shards']])
PY
done
