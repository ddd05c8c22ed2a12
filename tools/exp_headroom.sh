set -u
python bench.py --steps 5 --warmup 2 > gpurun_out/r3_bench_ns.json 2> gpurun_out/r3_bench_ns.err || echo FAILED ns
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_ns.json')); print('ns', d['value'], d['ms_per_step'], d['host_to_host'], d['cpu_baseline']['value'], d['roofline']['frac'])"
LGMI_MEM_BUDGET_MB=60000 python bench.py --workload headroom_dense_100kx200k --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > gpurun_out/r3_bench_headroom100k_60GB.json 2> gpurun_out/r3_bench_headroom100k_60GB.err || echo FAILED h100
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_headroom100k_60GB.json')); print('h100 60GB', d['value'], d['ms_per_step'], d['n_seq_shards'], d['config']['emitted_pairs_job'])"
python bench.py --workload headroom_dense_150kx200k --steps 1 --warmup 0 --no-cpu-baseline --no-host-to-host > gpurun_out/r3_bench_headroom150k.json 2> gpurun_out/r3_bench_headroom150k.err || echo FAILED h150
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_headroom150k.json')); print('h150', d['value'], d['ms_per_step'], d['n_seq_shards'], d['config']['emitted_pairs_job'])"
