# step and stage times of the bench workloads with the product library (or LGMI_LIB)
set -u
for wl in ${WLS:-north_star_dense_50kx200k cfg5_dense_depthx4_S10000 cfg3_22x9091x45455 cfg2_dense_10kx50k}; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/c_$wl.json 2> gpurun_out/c_$wl.err || { echo "failed $wl"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/c_$wl.json')); print('$wl', round(d['ms_per_step'],1), {k:round(v,1) for k,v in d['stage_ms'].items() if k in ('ms_count','ms_emit','ms_perm_fast','ms_perm_general')})"
done
