# L1 (TCP) counters of k_perm_general on the default bench: how many L1 accesses a look-up instruction makes, how many go on to the L2
set -eu
ROOT=$PWD; OUT=$ROOT/gpurun_out/pmc_perm_l1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0 ${BENCH_ARGS:-}"
K="--kernel-include-regex k_perm_general"
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum $K --kernel-trace --output-format csv -d $OUT/a -- $B > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum $K --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.log
rocprofv3 --pmc TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_GATE_EN2_sum $K --kernel-trace --output-format csv -d $OUT/c -- $B > $OUT/c.json 2> $OUT/c.log || true
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum $K --kernel-trace --output-format csv -d $OUT/d -- $B > $OUT/d.json 2> $OUT/d.log || true
cd $ROOT
python3 - <<'P'
import csv,glob,collections
for tag in 'abcd':   # (round 3 had a fifth pass with TA_* counters that never returned: no counter list, log or dispatch record of it
                     #  was kept, so its cause cannot be established after the fact — profiles/README.md lists the TA block as
                     #  NOT COLLECTED on this pool; the TCP_TA_* / TCP_*_TA_* counters of passes b and c cover the TA-TCP interface)
    for f in glob.glob('gpurun_out/pmc_perm_l1/%s/**/*counter_collection.csv'%tag, recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_perm_general' in r['Kernel_Name']:
                acc[r['Counter_Name']]+=float(r['Counter_Value'])
        print(tag, dict(acc))
P
