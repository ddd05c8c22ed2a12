# timing-only ablations of the permutation kernels (results are wrong by construction): rebuilds the library
# with -DLGMI_PABL=mask (bits: perm.hip) and prints the stage times of the default bench
for a in "$@"; do
  make -C l-giremi_amd -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DLGMI_PABL=$a" -B lib/liblgmi.so > /dev/null 2>&1
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null > gpurun_out/ablp_$a.json
  python -c "import json; d=json.load(open('gpurun_out/ablp_$a.json')); print('PABL $a perm ms', d['stage_ms']['ms_perm'])"
done
make -C l-giremi_amd -s -B lib/liblgmi.so > /dev/null 2>&1
