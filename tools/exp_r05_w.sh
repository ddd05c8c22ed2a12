set -u
LGMI_FUZZ_SEEDS=400 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/parity_fuzz.txt; cat gpurun_out/parity_fuzz.txt
