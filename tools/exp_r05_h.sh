set -u
gcc -shared -fPIC -o /tmp/abort_bt.so tools/src/abort_bt.c
LD_PRELOAD=/tmp/abort_bt.so timeout -k 10 1100 python -m pytest tests -x -q -m gpu -s > gpurun_out/t_h.log 2>&1; echo rc=$?
grep -v "^  File\|Extension modules" gpurun_out/t_h.log | grep -v "^\[lgmi\|^\.$" | tail -80 | cut -c1-400
