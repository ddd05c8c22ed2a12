# one rocprofv3 --pmc pass (SQ counters) of a bench workload, summary per kernel.  bash tools/quick_sq.sh TAG [workload] [extra bench args]
set -eu
TAG=${1:-s}; WL=${2:-north_star_dense_50kx200k}; shift || true; shift || true
ROOT=$PWD; OUT=$ROOT/gpurun_out/qs_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0 "$@" > $OUT/sq.json 2> $OUT/sq.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 \
    --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0 "$@" > $OUT/sq2.json 2> $OUT/sq2.log || true
cd $ROOT
python3 tools/pmc_summary.py sq $WL $OUT/sq 0 0 $OUT/sq_summary.json
python3 - <<PY
import json
d=json.load(open('$OUT/sq_summary.json'))['$WL']
for k,e in sorted(d.items(), key=lambda kv:-kv[1]['ms']):
    if e['ms']<0.5: continue
    c=e['counters']
    print('%-18s %8.2f ms  valu %.3e issue4 %.3f lanes %.1f salu/valu %.2f vmem_rd %.3e lds %.3e waves %.3e' % (k, e['ms'], e.get('valu_insts',0), e.get('valu_issue_frac',0), e.get('active_lanes',0), c.get('SQ_INSTS_SALU',0)/max(c.get('SQ_INSTS_VALU',1),1), c.get('SQ_INSTS_VMEM_RD',0), c.get('SQ_INSTS_LDS',0), c.get('SQ_WAVES',0)))
PY
python3 tools/pmc_summary.py sq $WL $OUT/sq2 0 0 $OUT/sq2_summary.json || true
python3 - <<PY
import json
try:
    d=json.load(open('$OUT/sq2_summary.json'))['$WL']
    for k,e in sorted(d.items(), key=lambda kv:-kv[1]['ms']):
        if e['ms']<0.5: continue
        print(k, e['ms'], {a:'%.3e'%b for a,b in e['counters'].items()})
except Exception as ex: print('sq2 failed', ex)
PY
