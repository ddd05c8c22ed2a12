# host -> host time of the one-shot call against the number of row ranges the permutation stage is cut into
set -u
mkdir -p gpurun_out
for k in 1 2 4 6 8; do
  LGMI_PERM_CHUNKS=$k timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/chunks_$k.json 2> gpurun_out/chunks_$k.err || { echo "chunks $k failed"; tail -5 gpurun_out/chunks_$k.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/chunks_$k.json')); h=d['host_to_host']
print('chunks $k: h2h', round(h['ms'],1), 'ms  min', round(h['ms_min'],1), ' kernels', round(h['kernels_ms'],1), ' d2h_bytes %.3g' % h['d2h_bytes'], ' step', round(d['ms_per_step'],1))"
done
