# perm stage time of the default bench with ablation / variant libraries: bash tools/exp_libs.sh name...  (l-giremi_amd/build_abl/liblgmi_<name>.so)
set -u
for n in "$@"; do
  LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_$n.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/lib_$n.json 2> gpurun_out/lib_$n.err || { echo "failed $n"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/lib_$n.json')); print('$n perm_general ms', round(d['stage_ms']['ms_perm_general'],1), 'perm_fast', round(d['stage_ms']['ms_perm_fast'],1))"
done
