set -eu
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_tcp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --shuffles 0 --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0"
K="--kernel-include-regex k_count_mfma_fp4"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum $K --kernel-trace --output-format csv -d $OUT/a -- $B > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum $K --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY $K --kernel-trace --output-format csv -d $OUT/c -- $B > $OUT/c.json 2> $OUT/c.log
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F6F4 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD $K --kernel-trace --output-format csv -d $OUT/d -- $B > $OUT/d.json 2> $OUT/d.log || true
cd $ROOT
python3 - <<'P'
import csv,glob,collections
for tag in 'abcd':
    for f in glob.glob('gpurun_out/pmc_tcp/%s/**/*counter_collection.csv'%tag, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_count_mfma_fp4' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value'])
        print(tag, dict(acc))
P
