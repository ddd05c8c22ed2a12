set -u
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1; echo profile rc=$?; tail -5 gpurun_out/profile_round_r05.log
timeout -k 10 500 python tools/shard_times.py 1 2 4 8 2>&1 | tail -4; cp gpurun_out/shard_times.json gpurun_out/r05_shard_times_north_star.json
bash tools/bench_all.sh r05 2>&1 | tail -12
