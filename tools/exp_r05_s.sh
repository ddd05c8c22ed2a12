# the CLI's parent process under cProfile at -t 16 on the 8,000-gene input (main thread only: what the pipeline's parent does)
set -u
W=/tmp/lgmi_cli_e2e
timeout -k 10 600 python tools/cli_e2e.py --genes 8000 --reads 4000000 --build_only --workdir $W 2> gpurun_out/cli_prof_build.err
export PYTHONPATH=$PWD/l-giremi_amd
timeout -k 10 300 python - 2> gpurun_out/cli_prof.err <<PY
import cProfile, sys
import lgmi.cli as c
W = '$W'
cProfile.run("c.main(['-b', W + '/e2e.bam', '-c', 'chrS', '-o', W + '/prof', '--genome_fasta', W + '/e2e.fa', '--snp_bcf', W + '/e2e.vcf', '--mi_calculation_only', '--skip_strand_correction', '-t', '16', '--n_shuffles', '1000', '--timing_json', 'gpurun_out/cli_prof_timing.json'])", 'gpurun_out/cli_parent.prof')
PY
cat gpurun_out/cli_prof_timing.json; echo
python - <<'PY'
import pstats
s = pstats.Stats('gpurun_out/cli_parent.prof')
s.sort_stats('cumulative').print_stats(45)
s.sort_stats('tottime').print_stats(25)
PY
