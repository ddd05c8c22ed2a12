set -eu
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_fast; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-include-regex "k_perm_fast|k_perm_general|k_emit" --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.log
cd $ROOT
python3 - <<'P'
import csv,glob,collections
for f in glob.glob('gpurun_out/pmc_fast/b/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        print(k, dict(v), 'alive frac (WAVE_CYCLES*4/(BUSY/32*WAVES))', v['SQ_WAVE_CYCLES']*4/((v['SQ_BUSY_CYCLES']/32)*v['SQ_WAVES']) if v['SQ_WAVES'] else None)
P
