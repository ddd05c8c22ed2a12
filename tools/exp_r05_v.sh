set -u
W=/tmp/lgmi_cli_e2e
timeout -k 10 600 python tools/cli_e2e.py --genes 8000 --reads 4000000 --build_only --workdir $W 2> gpurun_out/cli_prof_build.err
export PYTHONPATH=$PWD/l-giremi_amd LGMI_TRACE_JOBS=1
for T in 16 32; do
timeout -k 10 300 python -m lgmi.cli -b $W/e2e.bam -c chrS -o $W/trace --genome_fasta $W/e2e.fa --snp_bcf $W/e2e.vcf --mi_calculation_only --skip_strand_correction -t $T --n_shuffles 1000 --timing_json gpurun_out/cli_trace_timing_$T.json 2> gpurun_out/cli_jobs_$T.err
cat gpurun_out/cli_trace_timing_$T.json; echo
grep "lgmi jobs" gpurun_out/cli_jobs_$T.err | head -70
done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
