python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "perm or exceed or shuffl" 2>&1 | tail -2
bash tools/exp_libs.sh c384 c256 c640
BENCH_ARGS="--workload cfg5_dense_depthx4_S10000" bash tools/exp_libs.sh c384 c256 c640
