# timing-only ablations of the permutation kernels (results are wrong by construction): rebuilds the library
# with -DLGMI_PABL=mask (bits: perm.hip) and prints the perm stage time of the default bench
# usage: bash tools/abl_perm.sh [-s SHUFFLES] mask...
S=1000
if [ "$1" = "-s" ]; then S=$2; shift 2; fi
for a in "$@"; do
  make -C l-giremi_amd -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DLGMI_PABL=$a" -B lib/liblgmi.so > /dev/null 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --shuffles $S 2>/dev/null > gpurun_out/ablp_$a.json
  python -c "import json; d=json.load(open('gpurun_out/ablp_$a.json')); print('PABL $a S=$S perm ms', d['stage_ms']['ms_perm'])"
done
make -C l-giremi_amd -s -B lib/liblgmi.so > /dev/null 2>&1
