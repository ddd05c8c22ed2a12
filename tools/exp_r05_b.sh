set -u
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/t_gpu_all.log 2>&1; echo rc=$?; tail -45 gpurun_out/t_gpu_all.log
