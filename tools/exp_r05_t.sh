# round-end refresh: the whole GPU suite, the rocprof passes, every workload's bench line, the shard times
set -u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=6 > gpurun_out/t_gpu_all.log 2>&1; echo rc=$?; grep -v "^  File\|Extension modules" gpurun_out/t_gpu_all.log | tail -12 | cut -c1-200
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1; echo profile rc=$?
bash tools/bench_all.sh r05 2>&1 | tail -10
timeout -k 10 600 python tools/shard_times.py 1 2 4 8 2> gpurun_out/shard_times.err   # -> gpurun_out/shard_times.json; echo shard rc=$?
