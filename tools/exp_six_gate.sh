# k_perm_six + k_perm_general by the gate of the six-cell path (LGMI_PERM_SIX_PTS, sixteenths of a chord per shuffle): bash tools/exp_six_gate.sh [workload] pts...
set -u
WL=$1; shift
ROOT=$PWD; cd /tmp && export TMPDIR=/tmp
for pts in "$@"; do
  rm -rf $ROOT/gpurun_out/gate_$pts
  LGMI_PERM_SIX_PTS=$pts rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/gate_$pts -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > $ROOT/gpurun_out/gate_$pts.json 2> $ROOT/gpurun_out/gate_$pts.err
  (cd $ROOT && python3 tools/pmc_summary.py stats gpurun_out/gate_$pts gpurun_out/gate_$pts.csv)
  echo "SIX_PTS=$pts: $(grep -E 'k_perm_six|k_perm_general' $ROOT/gpurun_out/gate_$pts.csv | awk -F, '{printf "%s %.2f ms; ", $1, $4/1e6}') six rows $(python3 -c "import json; d=json.load(open('$ROOT/gpurun_out/gate_$pts.json')); print(d['perm_roofline']['six_cell_exact_rows'], 'of', d['perm_roofline']['larger_rows'])")"
done
