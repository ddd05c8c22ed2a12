# count kernel: time and FETCH_SIZE / WRITE_SIZE by the number of x-tile rows that sweep the y tiles together (LGMI_XG; default 4):
# the "squarer concurrent tile set per XCD" question of VERDICT r3 item 4b.  bash tools/exp_xg_traffic.sh 2 4 8 16
set -u
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for xg in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $ROOT/gpurun_out/xg_${xg}_$c
    LGMI_XG=$xg rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/xg_${xg}_$c -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-to-host --shuffles 0 --steps 1 --warmup 0 > /dev/null 2> $ROOT/gpurun_out/xg_${xg}_$c.err
  done
  python3 - <<P
import csv,glob
def tot(c):
    v=0.0; t=0
    for f in glob.glob('$ROOT/gpurun_out/xg_${xg}_%s/**/*counter_collection.csv'%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_count_mfma_fp4' in r['Kernel_Name'] and r['Counter_Name']==c:
                v+=float(r['Counter_Value']); t=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    return v,t
f,t=tot('FETCH_SIZE'); w,_=tot('WRITE_SIZE')
print('LGMI_XG=$xg  k_count_mfma_fp4 %.2f ms  FETCH x2 %.1f GB  WRITE %.1f GB  total %.1f GB' % (t/1e6, 2*f*1024/1e9, w*1024/1e9, (2*f+w)*1024/1e9))
P
done
