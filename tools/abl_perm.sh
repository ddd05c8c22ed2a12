# timing-only ablations of the permutation kernels (results are wrong by construction): builds a SEPARATE library
# (l-giremi_amd/build_abl/liblgmi_abl.so, selected with LGMI_LIB) with -DLGMI_PABL=mask (bits: perm.hip; unknown or
# removed bits are a compile error) and prints the perm stage time of the default bench.  The product library
# lib/liblgmi.so is never touched, and the bench's stderr is kept next to its output.
# usage: bash tools/abl_perm.sh [-s SHUFFLES] mask...
set -u
S=1000
if [ "${1:-}" = "-s" ]; then S=$2; shift 2; fi
mkdir -p gpurun_out l-giremi_amd/build_abl
for a in "$@"; do
  if ! make -C l-giremi_amd -s -j8 BUILD=build_abl/obj_$a LIB=build_abl/liblgmi_abl.so EXTRA="-DLGMI_PABL=$a" > gpurun_out/ablp_$a.build.log 2>&1; then
    echo "PABL $a: build refused (see gpurun_out/ablp_$a.build.log)"; continue
  fi
  LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_abl.so timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --shuffles $S \
      > gpurun_out/ablp_$a.json 2> gpurun_out/ablp_$a.err || { echo "PABL $a: bench failed, stopping (gpurun_out/ablp_$a.err)"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/ablp_$a.json')); print('PABL $a S=$S perm ms', d['stage_ms']['ms_perm'])"
done
