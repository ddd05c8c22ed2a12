# footprint regime: k_perm_enum / k_perm_general times for several (shuffles, LGMI_PERM_ENUM_MAX): bash tools/exp_fp_enum.sh
set -u
# the footprint workload's batch is built by a pool of forked workers: build (and cache) it with a plain python call BEFORE
# any rocprofv3 line — under --pmc the profiler's library has initialised the GPU before bench.py starts, and a fork after
# that is what bench.py's own "before anything touches the GPU" rule forbids (advice r3)
python3 -c "import sys; sys.path[:0]=['$PWD','$PWD/l-giremi_amd']; from lgmi.synth import footprint_blocks; footprint_blocks(20000, seed=20250810, cache_dir='/tmp')"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in "1000 4096" "2048 8192" "2048 0" "1000 0" "64 4096"; do
  set -- $cfg
  rm -rf $ROOT/gpurun_out/fp_tr
  LGMI_PERM_ENUM_MAX=$2 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/fp_tr -- python3 $ROOT/bench.py --workload footprints_20k --no-cpu-baseline --no-host-to-host --steps 3 --warmup 1 --shuffles $1 > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py stats $ROOT/gpurun_out/fp_tr $ROOT/gpurun_out/fp_st.csv > /dev/null
  echo "S=$1 ENUM_MAX=$2: $(grep -E 'k_perm_(general|enum|fast)' $ROOT/gpurun_out/fp_st.csv | awk -F, '{printf "%s %.2f ms  ", substr($1,7,14), $4/1e6}')"
done
