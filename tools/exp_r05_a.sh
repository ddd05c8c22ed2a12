set -u
timeout -k 10 600 python -m pytest tests/test_gpu_compact.py tests/test_integration_stub.py -x -q -m gpu > gpurun_out/t_compact.log 2>&1; echo rc=$?; tail -15 gpurun_out/t_compact.log
LGMI_TRACE_HOST=1 bash tools/exp_perm_chunks.sh 2>&1 | tail -20
grep "run:uploaded" gpurun_out/chunks_4.err | tail -2
