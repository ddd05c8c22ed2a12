# occupancy / shuffle-count experiment of k_perm_general: perm stage time at LGMI_PERM_WPC one-wave workgroups per CU
# and two shuffle counts (the difference per 1000 shuffles is the loop, the rest is the per-row set-up)
set -u
mkdir -p gpurun_out
for w in ${WPCS:-16 12 8 4}; do for s in ${SS:-1000 2000}; do
  LGMI_PERM_WPC=$w timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 --shuffles $s > gpurun_out/wpc_${w}_$s.json 2> gpurun_out/wpc_${w}_$s.err || { echo "failed $w $s"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/wpc_${w}_$s.json')); print('WPC $w S $s perm_general ms', round(d['stage_ms']['ms_perm_general'],1), 'perm_fast', round(d['stage_ms']['ms_perm_fast'],1))"
done; done
