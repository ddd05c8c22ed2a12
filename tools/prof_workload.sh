# rocprofv3 kernel trace of one bench workload: bash tools/prof_workload.sh <workload> <tag>  -> gpurun_out/prof_<tag>/, prints the per-kernel table
set -eu
WL=$1; TAG=$2
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
# the host-built workloads fork a pool of numpy workers: build (and cache) the batch before the profiler is around
python3 -c "
import sys; sys.path.insert(0, '$ROOT'); sys.path.insert(0, '$ROOT/l-giremi_amd')
import bench
w = bench.WORKLOADS['$WL']
if w.get('regime') == 'footprints':
    from lgmi.synth import footprint_blocks
    footprint_blocks(w['n_footprints'], seed=20250810, cache_dir='/tmp')
"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 3 --warmup 1 > $OUT/trace.json 2> $OUT/trace.log
cd $ROOT
python3 tools/pmc_summary.py stats $OUT/trace $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
