set -eu
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_perm; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0"
K="--kernel-include-regex k_perm_general"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY $K --kernel-trace --output-format csv -d $OUT/a -- $B > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU $K --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.log || true
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS $K --kernel-trace --output-format csv -d $OUT/c -- $B > $OUT/c.json 2> $OUT/c.log || true
cd $ROOT
python3 - <<'P'
import csv,glob,collections
for tag in 'abc':
    for f in glob.glob('gpurun_out/pmc_perm/%s/**/*counter_collection.csv'%tag, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_perm_general' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value'])
        print(tag, dict(acc))
    for f in glob.glob('gpurun_out/pmc_perm/%s/**/*kernel_trace.csv'%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_perm_general' in r['Kernel_Name']:
                print(tag,'ms',(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
P
