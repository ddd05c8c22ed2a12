# footprint regime with variant libraries: bash tools/exp_fp_libs.sh name...   (l-giremi_amd/build_abl/liblgmi_<name>.so; "base" = the product)
set -u
for n in "$@"; do
  L=$PWD/l-giremi_amd/build_abl/liblgmi_$n.so; [ "$n" = base ] && L=$PWD/l-giremi_amd/lib/liblgmi.so
  LGMI_LIB=$L timeout -k 10 200 python bench.py --workload footprints_20k --no-cpu-baseline --no-host-to-host --steps 5 --warmup 2 > gpurun_out/fplib_$n.json 2> gpurun_out/fplib_$n.err || { echo "failed $n"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/fplib_$n.json')); print('$n step', round(d['ms_per_step'],2), 'perm_general+enum', round(d['stage_ms']['ms_perm_general'],2), 'perm_fast', round(d['stage_ms']['ms_perm_fast'],2))"
done
