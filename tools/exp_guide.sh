set -u
for g in 8 10 11; do for w in 16 12; do
  LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_g$g.so LGMI_PERM_WPC=$w timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/g_${g}_$w.json 2> gpurun_out/g_${g}_$w.err || { echo "failed $g $w"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/g_${g}_$w.json')); print('guide bits $g WPC $w perm_general ms', round(d['stage_ms']['ms_perm_general'],1))"
done; done
