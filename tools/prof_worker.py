"""One extraction worker's job under cProfile: the first N footprints of a CLI run, through lgmi.region._extract_chunk the
way a pool worker runs them (native site extraction + packing), in this process.
    python tools/prof_worker.py WORKDIR [N]      (WORKDIR: tools/cli_e2e.py --build_only)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'l-giremi_amd')]
import lgmi.region as region  # noqa: E402
from lgmi import cli  # noqa: E402


class _Stop(Exception):
    pass


def main():
    w = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    got = {}

    def capture(footprints, sam, genome, *a, **kw):
        got['fp'], got['kw'] = list(footprints), kw
        raise _Stop()

    region.regions_mismatch_analysis = capture
    try:
        cli.main(['-b', w + '/e2e.bam', '-c', 'chrS', '-o', '/tmp/prof_worker_out', '--genome_fasta', w + '/e2e.fa', '--snp_bcf', w + '/e2e.vcf',
                  '--mi_calculation_only', '--skip_strand_correction', '-t', '4'])
    except _Stop:
        pass
    kw, fps = got['kw'], got['fp']
    own = ('min_common_reads', 'n_shuffles', 'seed', 'engine', 'concat', 'threads', 'reopen', 'timing', 'group', 'removed_sink', 'pairs_sink')
    fk = {k: v for k, v in kw.items() if k not in own}
    region._extract_chunk((kw['reopen'], fps[:50], fk, True, True))                     # files open, code paths warm
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    region._extract_chunk((kw['reopen'], fps[50:50 + n], fk, True, True))
    pr.disable()
    dt = time.perf_counter() - t0
    print('%d footprints of %d: %.3f s = %.2f ms per footprint' % (n, len(fps), dt, 1000 * dt / n))
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)


if __name__ == '__main__':
    main()
