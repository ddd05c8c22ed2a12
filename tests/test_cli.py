"""N2 — the l-giremi-compatible command line on a synthetic BAM: own BGZF/BAM/FASTA/VCF readers
(pysam is not installed here) and the --mi_calculation_only outputs."""
import os

import numpy as np
import pandas as pd
import pytest

from fakes import FakeGenome, FakeSam, simulate_region


def write_inputs(tmp_path, regions):
    """regions: {contig: (reads, genome, snps)} -> bam, fasta, vcf paths"""
    from lgmi.io import BamWriter
    bam, fa, vcf = str(tmp_path / 'in.bam'), str(tmp_path / 'genome.fa'), str(tmp_path / 'snps.vcf')
    w = BamWriter(bam, [(c, len(g)) for c, (_r, g, _s) in regions.items()])
    with open(fa, 'w') as f, open(vcf, 'w') as v:
        v.write('##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n')
        for contig, (reads, genome, snps) in regions.items():
            f.write('>%s some description\n' % contig)
            for k in range(0, len(genome), 60):
                f.write(genome[k:k + 60] + '\n')
            for p in snps:
                v.write('%s\t%d\t.\t%s\tN\t.\t.\t.\n' % (contig, p + 1, genome[p].upper()))
            for r in reads:
                cig, prev = [], None
                for pos, _b in r._blocks:
                    if prev is not None and pos != prev + 1:
                        cig.append([3, pos - prev - 1])
                    if cig and cig[-1][0] == 0:
                        cig[-1][1] += 1
                    else:
                        cig.append([0, 1])
                    prev = pos
                w.write(contig, r.reference_start, r.query_name, r.is_reverse, [tuple(c) for c in cig],
                        ''.join(b for _p, b in r._blocks), r._cs)
    w.close()
    return bam, fa, vcf


def regions_fixture():
    out = {}
    for k, contig in enumerate(['chrA', 'chrB']):
        reads, genome, snps, _ = simulate_region(seed=200 + k, n_reads=70 + 30 * k)
        out[contig] = (reads, genome, snps)
    return out


def test_bam_roundtrip_and_pileup(tmp_path):
    from lgmi.io import BamReader, FastaReader, VcfReader
    regions = regions_fixture()
    bam, fa, vcf = write_inputs(tmp_path, regions)
    rd = BamReader(bam)
    assert rd.references == ['chrA', 'chrB']
    genome = FastaReader(fa)
    snps = VcfReader(vcf)
    for contig, (reads, seq, snp_pos) in regions.items():
        got = list(rd.fetch(contig))
        assert [r.query_name for r in got] == [r.query_name for r in reads]
        for g, e in zip(got, reads):
            assert (g.reference_start, g.reference_end, g.is_reverse, g.get_tag('cs')) == \
                   (e.reference_start, e.reference_end, e.is_reverse, e._cs)
            assert g.query_sequence == ''.join(b for _p, b in e._blocks)
        assert genome.fetch(contig, 10, 40) == seq[10:40]
        assert [r.start for r in snps.fetch(contig, 0, len(seq))] == snp_pos
        # pile-up: the same bases per column as the fake; intron columns carry '' for the spanning reads
        fake_cols = {c.pos: (c.get_query_names(), c.get_query_sequences()) for c in FakeSam(reads).pileup(contig, 0, len(seq))}
        for c in rd.pileup(contig=contig, start=0, stop=len(seq)):
            names, bases = c.get_query_names(), c.get_query_sequences()
            real = [(n, b) for n, b in zip(names, bases) if b != '']
            exp = list(zip(*fake_cols.get(c.pos, ([], []))))
            assert real == exp
    with pytest.raises(KeyError):
        got[0].get_tag('XX')


def test_footprints(tmp_path):
    from lgmi.cli import get_footprints
    from lgmi.io import BamReader
    regions = regions_fixture()
    bam, _fa, _vcf = write_inputs(tmp_path, regions)
    fps = get_footprints(BamReader(bam), ['chrA', 'chrZ', 'chrB'], 2)
    assert [f[0] for f in fps] == ['chrA', 'chrB']
    for (chrom, lo, hi, n), (reads, _g, _s) in zip(fps, regions.values()):
        assert lo == min(r.reference_start for r in reads) and hi == max(r.reference_end for r in reads)
        assert n == len(reads)


def test_cli_refuses_what_it_does_not_cover(tmp_path):
    from lgmi import cli
    with pytest.raises(SystemExit):
        cli.main(['-b', 'x.bam', '--genome_fasta', 'g.fa'])                        # GLM path
    with pytest.raises(SystemExit):
        cli.main(['-b', 'x.bam', '--genome_fasta', 'g.fa', '--mi_calculation_only'])   # strand correction


@pytest.mark.gpu
def test_cli_mi_calculation_only(tmp_path):
    import lgmi
    from lgmi import cli, region
    regions = regions_fixture()
    bam, fa, vcf = write_inputs(tmp_path, regions)
    prefix = str(tmp_path / 'out')
    cli.main(['-b', bam, '-c', 'chrA', 'chrB', '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--mi_calculation_only', '--skip_strand_correction', '--n_shuffles', '50', '--seed', '3'])
    mi = pd.read_table(prefix + '.mi.txt')
    removed = pd.read_table(prefix + '.removed.txt')
    strand = pd.read_table(prefix + '.strand.txt')
    assert list(strand.columns) == ['read_name', 'original_read_strand', 'corrected_read_strand'] and len(strand) == 0
    assert list(mi.columns) == ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi', 'p_perm']
    assert list(removed.columns) == ['chromosome', 'strand', 'pos', 'removed']
    # the same numbers as the region-level drop-in on the duck-typed fakes (CLI defaults: min_allele_ratio 0.05,
    # min_total_depth 2, min_common 6), footprint by footprint
    eng = lgmi.default_engine()
    frames = []
    for contig, (reads, genome, snps) in regions.items():
        lo, hi = min(r.reference_start for r in reads), max(r.reference_end for r in reads)
        frames.append(region.region_mismatch_analysis(contig, lo, hi, FakeSam(reads), FakeGenome(genome), snp_positions=snps,
                                                      min_allele_ratio=0.05, min_total_depth=2, min_common_reads=6, engine=eng))
    exp_mi = pd.concat([f[1] for f in frames])
    exp_removed = pd.concat([f[2] for f in frames])
    assert len(mi) == len(exp_mi) and len(mi) > 10
    assert mi.iloc[:, :6].values.tolist() == exp_mi.iloc[:, :6].values.tolist()
    assert np.allclose(mi['mi'].values, exp_mi['mi'].values, atol=1e-12)
    # the BAM pile-up (like pysam's) also has columns inside introns (reference skips), which the duck-typed
    # fake does not emit.  Through the reference's look-up quirks every such column becomes an empty site reported
    # as 'too few usable reads after filters', and a site dropped by the window filter next to an intron is
    # re-created by a later look-up and re-reported with that reason.  So: every site the fake run removes is removed
    # here too (same reason, or the empty-site reason), and every other reason seen here is the fake run's.
    empty = 'too few usable reads after filters'
    got = {(c, st, p): r for c, st, p, r in removed.values.tolist()}
    exp = {(c, st, p): r for c, st, p, r in exp_removed.values.tolist()}
    for k, r in exp.items():
        assert got.get(k) in (r, empty), (k, r, got.get(k))
    for k, r in got.items():
        if r != empty:
            assert exp.get(k) == r
    introns = {(c, p) for c in regions for a, b in ((300, 380), (620, 700)) for p in range(a, b)}
    assert {(c, p) for (c, _s, p) in set(got) - set(exp)} <= introns
