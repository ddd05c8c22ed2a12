# emit stage with k_emit held to at least 5 / 6 / 8 waves per SIMD (build_abl/liblgmi_emitN.so: -DLGMI_EMIT_WPS=N) against the product library
set -u
ROOT=$PWD
for lib in l-giremi_amd/lib/liblgmi.so build_abl/liblgmi_emit5.so build_abl/liblgmi_emit6.so build_abl/liblgmi_emit8.so l-giremi_amd/lib/liblgmi.so build_abl/liblgmi_emit5.so; do
  LGMI_LIB=$ROOT/$lib timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-to-host > gpurun_out/abx_$(basename $lib .so).json 2> gpurun_out/abx_$(basename $lib .so).err
  python -c "import json; d=json.load(open('gpurun_out/abx_$(basename $lib .so).json')); print('$lib', {k: round(v,2) for k,v in d['stage_ms'].items() if k in ('ms_total','ms_count','ms_emit','ms_perm')})"
done
