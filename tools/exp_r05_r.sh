# host -> host time of the one-shot call under a few settings (one box):  bash tools/exp_r05_r.sh [setting ...]
set -u
run() { # label, env...
  local n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/h2h_$n.json 2> gpurun_out/h2h_$n.err || { echo "$n failed"; return 1; }
  python -c "import json; d=json.load(open('gpurun_out/h2h_$n.json')); h=d['host_to_host']; print('$n', round(d['ms_per_step'],1), 'h2h', round(h['ms'],1), 'min', round(h['ms_min'],1), [round(x,1) for x in h['ms_calls_after_the_first']])"
}
if [ $# -gt 0 ]; then
  for s in "$@"; do run "$(echo $s | tr '=' '_')" $s || exit 1; done
else
  run base X=1 && run base2 X=1 && run perm3 LGMI_PERM_CHUNKS=3 && run perm5 LGMI_PERM_CHUNKS=5 && run up6 LGMI_UPLOAD_CHUNKS=6 && run up12 LGMI_UPLOAD_CHUNKS=12 && run sdma0 HSA_ENABLE_SDMA=0
fi
