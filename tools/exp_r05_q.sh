set -u
timeout -k 10 600 python -m pytest tests/test_gpu_compact.py tests/test_gpu_shard.py -x -q -m gpu 2>&1 | tail -3 && bash tools/exp_h2h_trace.sh
