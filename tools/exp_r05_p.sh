# the GPU suite with every pool block poisoned before use and glibc's heap checks on: a kernel (or host loop) that reads
# what it never wrote fails every time instead of once in a hundred runs
set -u
export LGMI_POOL_POISON=1 MALLOC_PERTURB_=165 MALLOC_CHECK_=3
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=5 -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/t_gpu_poison.log
cat gpurun_out/t_gpu_poison.log
