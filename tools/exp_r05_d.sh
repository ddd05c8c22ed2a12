set -u
for K in 8 16; do
LGMI_UPLOAD_CHUNKS=$K LGMI_TRACE_HOST=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pipe_$K.json 2> gpurun_out/pipe_$K.err; echo bench rc=$?
grep "lgmi pipe" gpurun_out/pipe_$K.err | tail -$K
grep "run:uploaded" gpurun_out/pipe_$K.err | tail -1
done
