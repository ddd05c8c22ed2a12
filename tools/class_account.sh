# Instruction-class account of the permutation kernels: dynamic counts per VALU class from the SQ
# counters (three rocprofv3 --pmc passes of one bench step), priced with the issue costs measured at >= 2 waves per SIMD
# (profiles/r03_ubench_isa.txt / r03_ubench_mix.txt) against the kernels' SIMD cycles.
#   bash tools/class_account.sh r04 [workload]       -> profiles/r04_pmc_class.json (+ gpurun_out/profiles_r04/)
set -eu
R=${1:-r04}; WL=${2:-north_star_dense_50kx200k}
ROOT=$PWD; OUT=$ROOT/gpurun_out/class_${R}_$WL
rm -rf $OUT; mkdir -p $OUT $ROOT/profiles $ROOT/gpurun_out/profiles_$R
if [ "$WL" = footprints_20k ]; then
python3 -c "import sys; sys.path[:0]=['$PWD','$PWD/l-giremi_amd']; from lgmi.synth import footprint_blocks; footprint_blocks(20000, seed=20250810, cache_dir='/tmp')"
fi
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-host-to-host --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
    --kernel-trace --output-format csv -d $OUT/a -- $B > $OUT/a.json 2> $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD \
    --kernel-trace --output-format csv -d $OUT/b -- $B > $OUT/b.json 2> $OUT/b.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d $OUT/c -- $B > $OUT/c.json 2> $OUT/c.log
cd $ROOT
python3 tools/class_account.py $WL $OUT profiles/${R}_pmc_class.json
cp profiles/${R}_pmc_class.json gpurun_out/profiles_$R/
