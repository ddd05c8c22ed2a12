set -u
W=/tmp/lgmi_cli_e2e
timeout -k 10 600 python tools/cli_e2e.py --genes 8000 --reads 4000000 --build_only --workdir $W 2> gpurun_out/cli_prof_build.err
timeout -k 10 300 python tools/prof_worker.py $W 1000 2> gpurun_out/prof_worker.err
