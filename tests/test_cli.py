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


def run_cli_and_compare(tmp_path, extra=()):
    """the CLI on a synthetic BAM (own BGZF/BAI reader) against what the REFERENCE's footprint_bulk_calculation
    computes for the same reads (tests/golden/cli.json: region_mismatch_analysis with the CLI's defaults, driven
    by tests/golden/gen_golden.py through a pysam-like pile-up) — .mi.txt and .removed.txt row for row"""
    from conftest import load_golden
    from lgmi import cli
    gold = load_golden('cli.json')['cases']
    regions = regions_fixture()
    bam, fa, vcf = write_inputs(tmp_path, regions)
    assert os.path.exists(bam + '.bai')
    prefix = str(tmp_path / 'out')
    cli.main(['-b', bam, '-c', 'chrA', 'chrB', '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--mi_calculation_only', '--skip_strand_correction'] + list(extra))
    return compare_cli_outputs(prefix)


def compare_cli_outputs(prefix):
    """PREFIX.mi.txt / .removed.txt / .strand.txt of a run on regions_fixture() against tests/golden/cli.json"""
    from conftest import load_golden
    gold = load_golden('cli.json')['cases']
    mi = pd.read_table(prefix + '.mi.txt')
    removed = pd.read_table(prefix + '.removed.txt')
    strand = pd.read_table(prefix + '.strand.txt')
    assert list(strand.columns) == ['read_name', 'original_read_strand', 'corrected_read_strand'] and len(strand) == 0
    assert list(mi.columns)[:7] == ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
    assert list(removed.columns) == ['chromosome', 'strand', 'pos', 'removed']
    exp_mi = [r for c in gold for r in c['pair_mi']['data']]
    exp_removed = [r for c in gold for r in c['removed']['data']]
    assert [c['contig'] for c in gold] == ['chrA', 'chrB'] and len(exp_mi) > 50
    assert mi.iloc[:, :6].values.tolist() == [r[:6] for r in exp_mi]
    assert np.allclose(mi['mi'].values, [r[6] for r in exp_mi], atol=1e-6, rtol=0)
    assert removed.values.tolist() == exp_removed
    return mi


def test_cli_host_side_matches_the_reference(tmp_path, monkeypatch):
    """CPU: everything but the MI kernels (BAM/BAI/FASTA/VCF reading, footprints, filters, typing, table assembly,
    output files); the MI block is the CPU oracle and the engine a stub — test-only, the product has no CPU path"""
    import lgmi.engine
    from lgmi import region
    from oracle import mi_oracle

    def oracle_blocks(regions, mc=5, n_shuffles=0, seed=0, engine=None):
        out = []
        for mm, chrom in regions:
            kept, means = mi_oracle.region_mi(mm, mc)
            out.append(([[chrom, s] + r for s in '+-' for r in kept[s]], {s: dict(means[s]) for s in '+-'}, None))
        return out

    class NoEngine:
        def __init__(self, device=None):
            pass

        def close(self):
            pass
    def oracle_table(regions, mc=5, n_shuffles=0, seed=0, engine=None, batch=None, site_base=0, codes_out=None):
        blocks = oracle_blocks(regions, mc)
        cols = ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
        return (pd.DataFrame.from_records([r for recs, _m, _p in blocks for r in recs], columns=cols),
                [m for _r, m, _p in blocks])
    monkeypatch.setattr(region, 'regions_pair_mi', oracle_blocks)
    monkeypatch.setattr(region, 'regions_pair_mi_table', oracle_table)
    monkeypatch.setattr(region, '_STRIP_READS', False)        # the oracle stand-in reads the read lists
    monkeypatch.setattr(lgmi.engine, 'Engine', NoEngine)
    run_cli_and_compare(tmp_path)
    # -t 2: the site extraction of the two footprints in a process pool (the reference's -t), same files out
    run_cli_and_compare(tmp_path, ['-t', '2'])


@pytest.mark.gpu
def test_cli_mi_calculation_only(tmp_path):
    mi = run_cli_and_compare(tmp_path, ['--n_shuffles', '50', '--seed', '3'])
    assert list(mi.columns)[7:] == ['p_perm']
    assert ((mi['p_perm'] >= 1 / 51 - 1e-12) & (mi['p_perm'] <= 1)).all()


# ---------------------------------------------------------------- BASELINE.json configs[0]: 500 sites x 2k reads through the CLI
def cfg1_inputs(tmp_path):
    from conftest import load_golden
    gold = load_golden('cli_cfg1.json')['case']
    reads, genome, snps, _ = simulate_region(**gold['sim'])
    bam, fa, vcf = write_inputs(tmp_path, {gold['contig']: (reads, genome, snps)})
    return gold, bam, fa, vcf


def check_cfg1(prefix, gold):
    mi = pd.read_table(prefix + '.mi.txt')
    removed = pd.read_table(prefix + '.removed.txt')
    exp_mi, exp_removed = gold['pair_mi']['data'], gold['removed']['data']
    assert len(exp_mi) > 5000 and gold['n_sites'] > 300
    assert mi.iloc[:, :6].values.tolist() == [r[:6] for r in exp_mi]
    assert np.allclose(mi['mi'].values, [r[6] for r in exp_mi], atol=1e-6, rtol=0)
    assert removed.values.tolist() == exp_removed
    return mi


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'cli_cfg1.json')),
                    reason='tests/golden/cli_cfg1.json not generated')
def test_cli_cfg1_host_side_matches_the_reference(tmp_path, monkeypatch):
    """the cfg1-sized footprint (2,000 reads, several hundred candidate sites) through the BAM reader, the filters and
    the vectorised packer; MI block = the C oracle on the packed blocks (test-only stub, the product has no CPU path)"""
    import lgmi.engine
    from lgmi import region
    from lgmi.pack import pack_blocks
    from oracle import c_oracle

    def oracle_blocks(regions, mc=5, n_shuffles=0, seed=0, engine=None):
        out = []
        for mm, chrom in regions:
            records, means = [], {'+': {}, '-': {}}
            for strand in '+-':
                blk = mm.get(strand, {})
                if len(blk) < 2:
                    continue
                pb = pack_blocks([blk])
                res = c_oracle.run(pb, min_common=mc, het_only=True)
                pos, names = pb.site_pos.tolist(), pb.type_names
                records.extend([chrom, strand, pos[i], names[i], pos[j], names[j], m]
                               for i, j, m in zip(res['row_i'].tolist(), res['row_j'].tolist(), res['row_mi'].tolist()))
                for s_ in np.nonzero(res['site_n_pairs'])[0].tolist():
                    means[strand][pos[s_]] = float(res['site_mean_mi'][s_])
            out.append((records, means, None))
        return out

    class NoEngine:
        def __init__(self, device=None):
            pass

        def close(self):
            pass
    def oracle_table(regions, mc=5, n_shuffles=0, seed=0, engine=None, batch=None, site_base=0, codes_out=None):
        blocks = oracle_blocks(regions, mc)
        cols = ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
        return (pd.DataFrame.from_records([r for recs, _m, _p in blocks for r in recs], columns=cols),
                [m for _r, m, _p in blocks])
    monkeypatch.setattr(region, 'regions_pair_mi', oracle_blocks)
    monkeypatch.setattr(region, 'regions_pair_mi_table', oracle_table)
    monkeypatch.setattr(region, '_STRIP_READS', False)
    monkeypatch.setattr(lgmi.engine, 'Engine', NoEngine)
    gold, bam, fa, vcf = cfg1_inputs(tmp_path)
    from lgmi import cli
    prefix = str(tmp_path / 'cfg1')
    cli.main(['-b', bam, '-c', gold['contig'], '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--mi_calculation_only', '--skip_strand_correction'])
    check_cfg1(prefix, gold)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(__file__), 'golden', 'cli_cfg1.json')),
                    reason='tests/golden/cli_cfg1.json not generated')
def test_cli_cfg1_mi_calculation_only(tmp_path):
    from lgmi import cli
    gold, bam, fa, vcf = cfg1_inputs(tmp_path)
    prefix = str(tmp_path / 'cfg1')
    cli.main(['-b', bam, '-c', gold['contig'], '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--mi_calculation_only', '--skip_strand_correction', '--n_shuffles', '100', '--seed', '5'])
    mi = check_cfg1(prefix, gold)
    assert ((mi['p_perm'] >= 1 / 101 - 1e-12) & (mi['p_perm'] <= 1)).all()          # (read back from text)


def _compare_mip_table(path, gold):
    """PREFIX.mismatch_mip.txt against the reference's mismatch table + mip (tests/golden/cli_mip.json: the reference's
    region_mismatch_analysis and stat.ecdf, put together as script/giremi.py:415-429 does)"""
    got = pd.read_table(path, keep_default_na=False, na_values=[''])
    exp = pd.DataFrame(gold['data'], columns=gold['columns'])
    assert list(got.columns) == gold['columns'] and len(got) == len(exp) > 0
    for c in gold['columns']:
        if c in ('ratio', 'allelic_ratio_diff', 'mean_mi', 'mip'):
            a, b = got[c].to_numpy(np.float64), exp[c].to_numpy(np.float64)
            assert (np.isnan(a) == np.isnan(b)).all(), c
            bad = np.nonzero(~np.isclose(a, b, atol=1e-6, rtol=0, equal_nan=True))[0]
            if c == 'mip' and len(bad):
                # mip is a RANK among the het-SNP means: a row may differ from the reference by steps of 1/n only where
                # het-SNP means lie within the MI tolerance of its own mean_mi (here: 1e-9) and so may swap ranks
                het = exp.loc[(exp['type'] == 'het_snp') & exp['mean_mi'].notna(), 'mean_mi'].to_numpy(np.float64)
                for k in bad:
                    near = int((np.abs(het - exp['mean_mi'][k]) <= 1e-9).sum())
                    assert near >= 1 and abs(a[k] - b[k]) <= (near + 0.5) / len(het), (int(k), a[k], b[k], near)
                assert len(bad) <= 0.05 * len(a)
            else:
                assert len(bad) == 0, (c, bad[:5])
        else:
            assert got[c].astype(str).tolist() == exp[c].astype(str).tolist(), c
    # mip is the reference's own arithmetic (numpy linspace values picked by rank, stat.py:19-27; lgmi_ecdf restates it):
    # the written column equals the reference's, written the same way (pandas prints 16 significant digits), exactly —
    # unless two mean_mi that differ within the MI tolerance swap ranks
    import io
    buf = io.StringIO()
    exp[['mip']].to_csv(buf, sep='\t', index=False)
    want = pd.read_table(io.StringIO(buf.getvalue()))['mip'].to_numpy(np.float64)
    m = ~np.isnan(want)
    assert m.any() and np.mean(got['mip'].to_numpy(np.float64)[m] == want[m]) > 0.94


@pytest.mark.gpu
def test_cli_mip_table(tmp_path):
    """--mip_table: the reference's own p-value end to end — mean MI per site from the GPU, ECDF of the het-SNP means
    (lgmi_ecdf), the table the reference hands to its GLM"""
    from conftest import load_golden
    from lgmi import cli
    gold = load_golden('cli_mip.json')['cases']['cli']['mismatch_mip']
    bam, fa, vcf = write_inputs(tmp_path, regions_fixture())
    prefix = str(tmp_path / 'out')
    cli.main(['-b', bam, '-c', 'chrA', 'chrB', '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--skip_strand_correction', '--mip_table'])                       # without --mi_calculation_only: allowed with --mip_table
    compare_cli_outputs(prefix)
    _compare_mip_table(prefix + '.mismatch_mip.txt', gold)


@pytest.mark.gpu
def test_cli_cfg1_mip_table(tmp_path):
    from conftest import load_golden
    cases = load_golden('cli_mip.json')['cases']
    if 'cfg1' not in cases:
        pytest.skip('cfg1 part of tests/golden/cli_mip.json not generated')
    from lgmi import cli
    gold, bam, fa, vcf = cfg1_inputs(tmp_path)
    prefix = str(tmp_path / 'cfg1')
    cli.main(['-b', bam, '-c', gold['contig'], '-o', prefix, '--genome_fasta', fa, '--snp_bcf', vcf,
              '--mi_calculation_only', '--skip_strand_correction', '--mip_table'])
    _compare_mip_table(prefix + '.mismatch_mip.txt', cases['cfg1']['mismatch_mip'])


def test_removed_table_writer_matches_pandas(tmp_path):
    """cli._write_removed (pyarrow's CSV writer when present) writes what DataFrame.to_csv(sep='\\t', index=False) writes"""
    import filecmp
    from lgmi.cli import _write_removed
    n = 5000
    k = np.arange(n)
    df = pd.DataFrame({'chromosome': np.where(k % 7, 'chr1', 'chrUn_random').astype(object),
                       'strand': np.where(k % 2, '+', '-').astype(object), 'pos': k.astype(np.int64) * 3 - 5,
                       'removed': np.array(['too few usable reads after filters', 'in homopoly regions'], dtype=object)[k % 2]})
    cases = [(df, True), (df, False), (df.iloc[:0], True)]
    # what regions_mismatch_analysis(concat=True) hands the CLI: the string columns as categoricals
    as_cat = df.assign(chromosome=pd.Categorical(df['chromosome']), strand=pd.Categorical(df['strand'], categories=['+', '-']),
                       removed=pd.Categorical(df['removed']))
    cases.append((as_cat, True))
    odd = df.iloc[:10].copy()
    odd.loc[odd.index[3], 'removed'] = 'has\ttab'                     # pandas quotes it: the writer must fall back
    cases.append((odd, True))
    for frame, header in cases:
        a, b = str(tmp_path / 'a.txt'), str(tmp_path / 'b.txt')
        frame.to_csv(a, sep='\t', index=False, header=header)
        _write_removed(frame, b, header=header)
        assert filecmp.cmp(a, b, shallow=False)


def test_pair_table_written_in_parts_matches_one_to_csv(tmp_path):
    """cli._PairsWriter: the chunks of a pipelined run appended one by one give the bytes of one to_csv over the whole table;
    a run that sent no chunk writes the whole table at close; abort leaves nothing behind"""
    import filecmp
    from lgmi.cli import _PairsWriter
    rng = np.random.default_rng(3)
    n = 3000
    df = pd.DataFrame({'chromosome': np.array(['chr1', 'chr2'], dtype=object)[rng.integers(0, 2, n)],
                       'site1_pos': rng.integers(1, 10 ** 8, n), 'site2_pos': rng.integers(1, 10 ** 8, n),
                       'mi': rng.random(n) * np.where(rng.random(n) < 0.1, 0.0, 1.0), 'n': rng.integers(6, 2000, n),
                       'p_perm': rng.integers(1, 1002, n) / 1001.0})
    want = str(tmp_path / 'want.txt')
    df.to_csv(want, sep='\t', index=False)
    w = _PairsWriter(str(tmp_path / 'parts.txt'))
    w.FLUSH_ROWS = 900                                    # several flushes, one of them of gathered chunks, and a rest at close
    for a, b in ((0, 1), (1, 500), (500, 1700), (1700, 2800), (2800, n)):
        w(df.iloc[a:b])
    assert not os.path.exists(w.final)
    w.close(df)
    assert filecmp.cmp(want, w.final, shallow=False) and not os.path.exists(w.path)
    w = _PairsWriter(str(tmp_path / 'whole.txt'))
    w.close(df)
    assert filecmp.cmp(want, w.final, shallow=False)
    w = _PairsWriter(str(tmp_path / 'gone.txt'))
    w(df.iloc[:5])
    w.abort()
    assert not os.path.exists(w.path) and not os.path.exists(w.final)
    # chunks that bring dictionary codes for their string columns are written natively (lgio_write_table): the same bytes,
    # also mixed with chunks that do not, and with a name pandas would quote (that chunk falls back)
    types = np.array(['het_snp', 'mismatch', 'edit'], dtype=object)
    chroms = np.array(['chr1', 'chr\t2'], dtype=object)
    cc, t1, t2 = rng.integers(0, 2, n), rng.integers(0, 3, n), rng.integers(0, 3, n)
    mi = df['mi'].to_numpy().copy()
    mi[::97] = np.nan
    mi[5] = 1e-7
    mi[6] = 123456789012345680.0
    full = pd.DataFrame({'chromosome': np.array(['chr1', 'chr2'], dtype=object)[cc], 'strand': np.array(['+', '-'], dtype=object)[cc ^ 1],
                         'site1_pos': df['site1_pos'], 'site1_type': types[t1], 'site2_pos': df['site2_pos'], 'site2_type': types[t2],
                         'mi': mi, 'p_perm': df['p_perm']})

    def codes_of(sl, names=('chr1', 'chr2')):
        return {'chromosome': (cc[sl].astype(np.int32), list(names)), 'strand': ((cc[sl] ^ 1).astype(np.int32), ['+', '-']),
                'site1_type': (t1[sl].astype(np.int32), list(types)), 'site2_type': (t2[sl].astype(np.int32), list(types))}

    full.to_csv(want, sep='\t', index=False)
    w = _PairsWriter(str(tmp_path / 'native.txt'))
    w.FLUSH_ROWS = 900
    for k, (a, b) in enumerate(((0, 700), (700, 800), (800, 2000), (2000, 2001), (2001, n))):
        sl = slice(a, b)
        w(full.iloc[sl], codes_of(sl) if k != 1 else None)                      # (the second chunk without codes: pandas, in order)
    w.close(full)
    assert filecmp.cmp(want, w.final, shallow=False)
    odd = full.iloc[:50].copy()
    odd['chromosome'] = chroms[cc[:50]]
    odd.to_csv(want, sep='\t', index=False)
    w = _PairsWriter(str(tmp_path / 'quoted.txt'))
    w(odd, codes_of(slice(0, 50), names=tuple(chroms)))
    w.close(odd)
    assert filecmp.cmp(want, w.final, shallow=False)
