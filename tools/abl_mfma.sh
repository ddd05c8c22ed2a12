# timing-only ablations of k_count_mfma_fp4 (results are wrong by construction): builds a SEPARATE library
# (l-giremi_amd/build_abl/liblgmi_abl.so, selected with LGMI_LIB) with -DLGMI_ABL=n; lib/liblgmi.so is never touched
set -u
mkdir -p gpurun_out l-giremi_amd/build_abl
for a in ${@:-0 1 2 3}; do
  make -C l-giremi_amd -s -j8 BUILD=build_abl/objm_$a LIB=build_abl/liblgmi_abl.so EXTRA="-DLGMI_ABL=$a" > gpurun_out/ablm_$a.build.log 2>&1 || { echo "ABL $a: build failed"; continue; }
  LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_abl.so timeout -k 10 200 python bench.py --shuffles 0 --no-cpu-baseline --steps 2 \
      > gpurun_out/ablm_$a.json 2> gpurun_out/ablm_$a.err || { echo "ABL $a: bench failed, stopping"; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/ablm_$a.json')); print('ABL $a', d['stage_ms']['ms_count'])"
done
