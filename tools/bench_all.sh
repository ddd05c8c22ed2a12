# every workload's bench line, as committed under profiles/rNN_bench_*.json.  On an MI355X box:  bash tools/bench_all.sh r02
set -u
R=${1:-r02}; OUT=gpurun_out/bench_$R; mkdir -p $OUT
run() { # name, args...
  local n=$1; shift
  timeout -k 10 500 python bench.py "$@" > $OUT/${R}_bench_$n.json 2> $OUT/$n.err || { echo "$n: bench failed ($OUT/$n.err)"; return 1; }
  python -c "import json; d=json.load(open('$OUT/${R}_bench_$n.json')); print('$n', round(d['ms_per_step'],2), 'ms/step', '%.3g' % d['value'], d['unit'], 'roofline', round(d['roofline']['frac'],3))"
}
run ns && run ns_s0 --shuffles 0 --no-cpu-baseline && run cfg2_dense_10kx50k --workload cfg2_dense_10kx50k && \
run cfg3_22x9091x45455 --workload cfg3_22x9091x45455 && run cfg5_dense_depthx4_S10000 --workload cfg5_dense_depthx4_S10000 && \
run north_star_banded_50kx200k --workload north_star_banded_50kx200k
# round 3: the footprint-shaped regime and the two headroom blocks (the 100k block under a 60 GB budget: sequential shards).
# One warm-up step: the first call of the 150k block allocates ~250 GB, which takes anything from 0 to 5.5 s on the same box.
run footprints_20k --workload footprints_20k && \
LGMI_MEM_BUDGET_MB=60000 run headroom_dense_100kx200k_60GB --workload headroom_dense_100kx200k --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host && \
run headroom_dense_150kx200k --workload headroom_dense_150kx200k --steps 1 --warmup 1 --no-cpu-baseline --no-host-to-host
