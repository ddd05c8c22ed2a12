set -u
LGMI_FUZZ_SEEDS=400 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/parity_fuzz.txt; cat gpurun_out/parity_fuzz.txt
# how many of the first 80 cases went through the pipelined upload / cut themselves into sequential shards (host trace lines)
LGMI_TRACE_HOST=1 LGMI_FUZZ_SEEDS=80 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider -s > gpurun_out/parity_fuzz_trace.txt 2>&1
echo "calls with upload pieces: $(grep -c 'piece=' gpurun_out/parity_fuzz_trace.txt)   host trace lines: $(grep -c 'lgmi host' gpurun_out/parity_fuzz_trace.txt)" | tee -a gpurun_out/parity_fuzz.txt
tail -2 gpurun_out/parity_fuzz_trace.txt
