set -u
timeout -k 10 900 python tools/cli_e2e.py --genes 8000 --reads 4000000 --threads 16 8 --n_shuffles 1000 --workdir /tmp/lgmi_cli_e2e 2> gpurun_out/cli_e2e.err | tee gpurun_out/cli_e2e_r05.txt
tail -3 gpurun_out/cli_e2e.err
timeout -k 10 600 python -m pytest tests/test_cli.py tests/test_region.py tests/test_gpu_gather2.py -x -q -m gpu 2>&1 | tail -3
