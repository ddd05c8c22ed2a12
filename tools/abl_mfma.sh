# timing-only ablations of k_count_mfma (results are wrong by construction): rebuilds the library with -DLGMI_ABL=n
for a in ${@:-0 1 2 3}; do
  make -C l-giremi_amd -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DLGMI_ABL=$a" -B lib/liblgmi.so > /dev/null 2>&1
  timeout -k 10 200 python bench.py --shuffles 0 --no-cpu-baseline --steps 2 2>/dev/null > gpurun_out/ablm_$a.json
  python -c "import json; d=json.load(open('gpurun_out/ablm_$a.json')); print('ABL $a', d['stage_ms']['ms_count'])"
done
make -C l-giremi_amd -s -B lib/liblgmi.so > /dev/null 2>&1
