set -u
ROOT=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py tests/test_gpu_gather2.py tests/test_gpu_shard.py tests/test_gpu_parity.py tests/test_cli.py tests/test_splice.py tests/test_region.py -x -q -m gpu > gpurun_out/t_e.log 2>&1; echo rc=$?; tail -6 gpurun_out/t_e.log
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof_e
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_e -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-to-host > $ROOT/gpurun_out/prof_e.json 2> $ROOT/gpurun_out/prof_e.log
cd $ROOT
python3 tools/pmc_summary.py stats gpurun_out/prof_e gpurun_out/prof_e_stats.csv && head -12 gpurun_out/prof_e_stats.csv | cut -c1-60,180-400
