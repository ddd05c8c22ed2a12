# per-kernel times of the default bench with each of several pre-built library variants: bash tools/lib_variants.sh LIB...
set -u
cd /tmp && export TMPDIR=/tmp && cd $OLDPWD
for lib in "$@"; do
  tag=$(basename $lib .so)
  LGMI_LIB=$PWD/$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_$tag -- python3 bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/var_$tag.json 2> gpurun_out/var_$tag.err
  python3 tools/pmc_summary.py stats gpurun_out/var_$tag gpurun_out/var_$tag.csv
  echo "$tag: $(grep -E "k_perm_six|k_perm_fast|k_perm_general|k_emit" gpurun_out/var_$tag.csv | awk -F, '{printf "%s %.2f ms; ", $1, $4/1e6}')"
done
