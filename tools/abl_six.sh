# timing-only ablations of k_perm_six (results wrong by construction): the variants are cross-compiled HERE into
# l-giremi_amd/build_abl/liblgmi_six$M.so (bash tools/abl_six.sh build 1 2 3), then timed on the GPU box
# (bash tools/abl_six.sh run 0 1 2 3) with LGMI_LIB; the product library is never touched.
set -u
MODE=$1; shift
for a in "$@"; do
  if [ "$MODE" = build ]; then
    make -C l-giremi_amd -s -j8 BUILD=build_abl/obj_six$a LIB=build_abl/liblgmi_six$a.so EXTRA="-DLGMI_SIXABL=$a" || exit 1
  else
    LGMI_LIB=$PWD/l-giremi_amd/build_abl/liblgmi_six$a.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/six_abl_$a -- python3 bench.py --no-cpu-baseline --no-host-to-host --steps 2 --warmup 1 > gpurun_out/six_abl_$a.json 2> gpurun_out/six_abl_$a.err
    python3 tools/pmc_summary.py stats gpurun_out/six_abl_$a gpurun_out/six_abl_$a.csv
    echo "SIXABL $a: $(grep k_perm_six gpurun_out/six_abl_$a.csv | cut -d, -f1,4)"
  fi
done
