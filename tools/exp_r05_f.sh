# A/B on one box: round-4 library vs this round's (pseudo rows next to their x rows) — every shard of the north-star cut 8 ways;
# and k_perm_fast with 32-value units (timing only: -DLGMI_UNIT=32 changes the specification's unit cut)
set -u
ROOT=$PWD
( cd build_abl/r4 && timeout -k 10 400 python tools/shard_times.py 1 8 2>&1 | tail -2 ) | sed 's/^/r4:  /'
timeout -k 10 400 python tools/shard_times.py 1 8 2>&1 | tail -2 | sed 's/^/r5:  /'
cp gpurun_out/shard_times.json gpurun_out/shard_times_r5.json; cp build_abl/r4/gpurun_out/shard_times.json gpurun_out/shard_times_r4.json
for lib in l-giremi_amd/lib/liblgmi.so build_abl/liblgmi_u32.so; do
  LGMI_LIB=$ROOT/$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > gpurun_out/ab_$(basename $lib .so).json 2> gpurun_out/ab_$(basename $lib .so).err
  python -c "import json; d=json.load(open('gpurun_out/ab_$(basename $lib .so).json')); print('$lib', {k: round(v,2) for k,v in d['stage_ms'].items()})"
done
