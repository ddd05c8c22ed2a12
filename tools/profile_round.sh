# the rocprofv3 passes behind profiles/rNN_*: kernel trace, then the counters in their own runs (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc is never combined with other trace domains).  Run from the repo root on an
# MI355X box:   bash tools/profile_round.sh r02 [workload]
set -eu
R=${1:-r02}; WL=${2:-north_star_dense_50kx200k}
ROOT=$PWD; OUT=$ROOT/gpurun_out/prof_${R}_$WL
rm -rf $OUT; mkdir -p $OUT $ROOT/profiles $ROOT/gpurun_out/profiles_$R
if [ "$WL" = footprints_20k ]; then
# the footprint workload's batch is built by a pool of forked workers: build (and cache) it with a plain python call BEFORE
# any rocprofv3 line — under --pmc the profiler's library has initialised the GPU before bench.py starts, and a fork after
# that is what bench.py's own "before anything touches the GPU" rule forbids (advice r3)
python3 -c "import sys; sys.path[:0]=['$PWD','$PWD/l-giremi_amd']; from lgmi.synth import footprint_blocks; footprint_blocks(20000, seed=20250810, cache_dir='/tmp')"
fi
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 3 --warmup 1 > $OUT/trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $B --steps 1 --warmup 0 > $OUT/fetch.json 2> $OUT/fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $B --steps 1 --warmup 0 > $OUT/write.json 2> $OUT/write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $OUT/sq -- $B --steps 1 --warmup 0 > $OUT/sq.json 2> $OUT/sq.log
cd $ROOT
python3 tools/pmc_summary.py stats $OUT/trace profiles/${R}_${WL}_kernel_stats.csv
python3 tools/pmc_summary.py count $WL $OUT/fetch $OUT/write profiles/${R}_pmc_k_count.json
DRAWS=$(python3 -c "import json; d=json.load(open('$OUT/sq.json')); print(d['perm_roofline']['table_draws_upper'])")
ROWS=$(python3 -c "import json; d=json.load(open('$OUT/sq.json')); print(d['perm_roofline']['two_by_two_rows'])")
python3 tools/pmc_summary.py sq $WL $OUT/sq $DRAWS $ROWS profiles/${R}_pmc_sq_perm.json
cp $OUT/trace.json profiles/${R}_bench_under_rocprof_${WL}.json
# gpurun merges only gpurun_out/ back: the summaries travel there (copy them into profiles/ and commit)
cp profiles/${R}_${WL}_kernel_stats.csv profiles/${R}_pmc_k_count.json profiles/${R}_pmc_sq_perm.json profiles/${R}_bench_under_rocprof_${WL}.json gpurun_out/profiles_$R/
echo "profiles written:"; ls -la profiles | grep $R
