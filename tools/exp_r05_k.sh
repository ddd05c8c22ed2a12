set -u
ROOT=$PWD
( cd build_abl/r4 && timeout -k 10 300 python tools/shard_times.py 1 > /dev/null 2>&1; python -c "import json; d=json.load(open('gpurun_out/shard_times.json')); print('r4  ', d['1']['shards'][0])" )
for lib in l-giremi_amd/lib/liblgmi.so build_abl/liblgmi_nopf.so; do
  LGMI_LIB=$ROOT/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > gpurun_out/ab2_$(basename $lib .so).json 2> gpurun_out/ab2_$(basename $lib .so).err
  python -c "import json; d=json.load(open('gpurun_out/ab2_$(basename $lib .so).json')); print('$lib', {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('ms_total','ms_count','ms_emit','ms_perm')})"
done
