# one rocprofv3 kernel-trace pass of a bench workload; prints the per-kernel summary.  bash tools/quick_trace.sh TAG [workload] [extra bench args]
set -eu
TAG=${1:-t}; WL=${2:-north_star_dense_50kx200k}; shift || true; shift || true
ROOT=$PWD; OUT=$ROOT/gpurun_out/qt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 3 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/trace.log
cd $ROOT
python3 tools/pmc_summary.py stats $OUT/trace $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
