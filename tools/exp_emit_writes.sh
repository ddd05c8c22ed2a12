# WRITE_SIZE / FETCH_SIZE and time of k_emit<2> with parts of its stores compiled out (libraries built with -DLGMI_EABL_W=mask):
# what the 24 GB it writes per north-star launch are.  bash tools/exp_emit_writes.sh LIB...
set -u
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  rm -rf $ROOT/gpurun_out/ew_$tag
  LGMI_LIB=$ROOT/$lib rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $ROOT/gpurun_out/ew_$tag -- python3 $ROOT/bench.py --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0 > /dev/null 2> $ROOT/gpurun_out/ew_$tag.err
  python3 - <<P
import csv,glob
for f in glob.glob('$ROOT/gpurun_out/ew_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_emit<2>' in r['Kernel_Name'] and r['Counter_Name']=='WRITE_SIZE':
            print('$tag  k_emit<2> %.2f ms  WRITE %.2f GB' % ((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, float(r['Counter_Value'])*1024/1e9))
P
done
