# perm parity subset, then the perm stage times of the product library on north-star / cfg5 / cfg3:  bash tools/exp_cur.sh
set -u
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "perm or exceed or shuffl or exact" > gpurun_out/t_cur.txt 2>&1; tail -2 gpurun_out/t_cur.txt
for wl in north_star_dense_50kx200k cfg5_dense_depthx4_S10000 cfg3_22x9091x45455 cfg2_dense_10kx50k; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/cur_$wl.json 2> gpurun_out/cur_$wl.err || { echo "failed $wl"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/cur_$wl.json')); print('$wl step', round(d['ms_per_step'],1), 'perm_general', round(d['stage_ms']['ms_perm_general'],1), 'perm_fast', round(d['stage_ms']['ms_perm_fast'],1))"
done
for wl in footprints_20k; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-to-host --steps 3 --warmup 1 > gpurun_out/cur_$wl.json 2> gpurun_out/cur_$wl.err || { echo "failed $wl"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/cur_$wl.json')); print('$wl step', round(d['ms_per_step'],2), 'perm_general', round(d['stage_ms']['ms_perm_general'],2), 'perm_fast', round(d['stage_ms']['ms_perm_fast'],2))"
done
