"""``l-giremi``-compatible command line for the ``--mi_calculation_only`` path.

Same flags and the same three outputs as the reference's entry point
(src/giremi/script/giremi.py:140-321 flags, :396-409 outputs):

    PREFIX.strand.txt    read_name, original_read_strand, corrected_read_strand
    PREFIX.mi.txt        chromosome, strand, site1_pos, site1_type, site2_pos, site2_type, mi
    PREFIX.removed.txt   chromosome, strand, pos, removed

What differs by design: the reference maps ``footprint_bulk_calculation`` over a process pool and calls
the MI step inside the workers (:375-380) — a HIP context does not survive that fork and per-footprint
launches are tiny — so here the site extraction runs footprint by footprint on the host and the MI blocks
of ALL footprints go to the GPU as one batch (``lgmi.region.regions_mismatch_analysis``).

Not covered: the GLM scoring after the MI step (stat.py:32-143) and the GTF-based strand correction
(strand.py): run with ``--mi_calculation_only --skip_strand_correction``; anything else exits with a
message instead of producing partial output.  New flags: ``--n_shuffles``/``--seed`` (permutation
p-value column ``p_perm``), ``--device``.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from collections import defaultdict

import numpy as np
import pandas as pd

from . import __version__
from .io import open_alignment, open_fasta, open_variants


def parse_args(argv=None):
    p = argparse.ArgumentParser(
        description='L-GIREMI site-pair MI step on MI355X (l-giremi compatible flags; --mi_calculation_only path)')
    p.add_argument('-b', '--bam_file', type=str, default=None, required=True,
                   help='input bam file, with cs tags, sorted and indexed')
    p.add_argument('-c', '--chromosomes', nargs='*', type=str,
                   default=['chr%s' % c for c in list(range(1, 23)) + ['X', 'Y']], help='chromosomes to be analyzed')
    p.add_argument('-o', '--output_prefix', type=str, default='out', help='prefix of output file')
    p.add_argument('-t', '--thread', type=int, default=1,
                   help='processes for the host-side site extraction (the MI step itself runs on the GPU, in one batch)')
    p.add_argument('--annotation_gtf', type=str, default=None)
    p.add_argument('--genome_fasta', type=str, default=None)
    p.add_argument('--homopoly_length', type=int, default=5)
    p.add_argument('--keep_non_spliced_read', action='store_true')
    p.add_argument('--max_het_snp_ratio', type=float, default=0.65)
    p.add_argument('--mi_calculation_only', action='store_true')
    p.add_argument('--mi_min_common_read', type=int, default=6)
    p.add_argument('--mi_p_threshold', type=float, default=0.05)
    p.add_argument('--min_allele_depth', type=int, default=3)
    p.add_argument('--min_allele_ratio', type=float, default=0.05)
    p.add_argument('--min_het_snp_ratio', type=float, default=0.35)
    p.add_argument('--min_total_depth', type=int, default=2)
    p.add_argument('--min_dist_from_splice', type=int, default=4)
    p.add_argument('--mismatch_window_size', type=int, default=100)
    p.add_argument('--max_window_mismatch', type=int, default=10)
    p.add_argument('--max_window_mismatch_type', type=int, default=3)
    p.add_argument('--mode', type=str, default='cs')
    p.add_argument('--model', type=str, default='lm')
    p.add_argument('--padding_exon', type=int, default=10)
    p.add_argument('--padding_gene', type=int, default=500)
    p.add_argument('--repeat_txt', type=str, default=None)
    p.add_argument('--skip_strand_correction', action='store_true')
    p.add_argument('--snp_bcf', type=str, default=None)
    p.add_argument('--n_shuffles', type=int, default=0, help='permutation shuffles per pair (0: no p_perm column)')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--device', type=int, default=None, help='GPU index (default: LGMI_DEVICE / LOCAL_RANK / 0)')
    p.add_argument('--timing_json', type=str, default=None, help='write the wall time of every stage of the run to this file')
    p.add_argument('--mip_table', action='store_true',
                   help='also write PREFIX.mismatch_mip.txt: the mismatch table with mean_mi and the reference\'s own p-value '
                        'column mip (ECDF of the het-SNP mean MI, script/giremi.py:415-429) — the table the reference hands to '
                        'its GLM, which is not part of lgmi; works with or without --mi_calculation_only')
    p.add_argument('--gpus', type=int, default=None,
                   help='ranks of a multi-GPU run (launch with python -m torch.distributed.run --nproc-per-node N -m lgmi.cli ... ; '
                        'default WORLD_SIZE): footprints are dealt to the ranks in contiguous, read-balanced runs, every rank '
                        'extracts and computes its own, the pair rows are gathered over RCCL onto rank 0, which writes')
    p.add_argument('--version', action='version', version='lgmi %s' % __version__)
    return p.parse_args(argv)


class _RemovedWriter:
    """the removed-site table written part by part, in footprint order, while the run is still extracting later footprints
    (regions_mismatch_analysis: removed_sink): the header with the first part, every further part appended — the bytes of
    one _write_removed over the whole table"""

    def __init__(self, path):
        # written under a temporary name and renamed when the run has succeeded: a run that fails half-way (advice r4: an
        # IndexError parity check, an out-of-memory kill) must not leave a truncated table that looks like a result
        self.final, self.path, self.parts = path, path + '.partial', 0
        self._q = self._thread = self._err = None
        if os.path.exists(self.final):
            os.remove(self.final)                          # (a stale table of an earlier run next to this run's other files)

    def __call__(self, df):
        self._drain()                                      # (parts are appended in the order they were handed over)
        _write_removed(df, self.path, header=(self.parts == 0), append=(self.parts > 0))
        self.parts += 1

    def write_arrays(self, arrays):
        """a chunk's removed sites as its extraction worker flattened them (lgmi.region._removed_arrays: names + codes):
        formatted by liblgmi_io (lgio_write_removed_table) — no DataFrame, no categoricals; 15 M rows of an 8,000-gene run
        were 1.3 s of the parent's time through pandas + pyarrow.  The call is queued to ONE writer thread (the native
        writer holds no interpreter lock: 450 MB of text leave the parent's critical path, 0.6 s of that run); what it
        raises comes out of the next call or of close()"""
        if self._thread is None:
            import queue
            import threading
            self._q = queue.Queue(maxsize=8)               # (bounded: a slow disk holds the pipeline back instead of the rows piling up)
            self._thread = threading.Thread(target=self._writer, name='lgmi-removed-writer', daemon=True)
            self._thread.start()
        if self._err is not None:
            self._drain()
        self._q.put(arrays)

    def _writer(self):
        while True:
            arrays = self._q.get()
            if arrays is None:
                return
            if self._err is None:
                try:
                    self._write_arrays_now(arrays)
                except BaseException as e:                  # noqa: BLE001 — handed to the thread that owns the run
                    self._err = e

    def _write_arrays_now(self, arrays):
        from .io import write_removed_table
        chroms, reasons, g_chrom, g_strand, g_pos, g_reason = arrays
        try:
            write_removed_table(self.path, chroms, reasons, g_chrom, g_strand, g_pos, g_reason, header=(self.parts == 0),
                                append=(self.parts > 0))
        except ValueError:                                  # a name pandas would quote: pandas writes it
            from .region import _removed_frame_from_arrays
            _write_removed(_removed_frame_from_arrays(arrays, {}, {}), self.path, header=(self.parts == 0), append=(self.parts > 0))
        self.parts += 1

    def _drain(self):
        if self._thread is not None:
            self._q.put(None)
            self._thread.join()
            self._thread = None
        if self._err is not None:
            e, self._err = self._err, None
            raise e

    def close(self):
        self._drain()
        if not self.parts:
            import pandas as pd
            _write_removed(pd.DataFrame({c: [] for c in ('chromosome', 'strand', 'pos', 'removed')}), self.path)
        os.replace(self.path, self.final)

    def abort(self):
        try:
            self._drain()
        except BaseException:                               # noqa: BLE001 — the run already failed; its own error is the one to show
            pass
        if os.path.exists(self.path):
            os.remove(self.path)


class _PairsWriter:
    """PREFIX.mi.txt written chunk by chunk while later footprints are still being extracted (regions_mismatch_analysis:
    pairs_sink) — the bytes of one df.to_csv(sep='\\t', index=False) over the whole table (every value is formatted on its
    own); under a temporary name until the run has succeeded"""

    FLUSH_ROWS = 32768            # one to_csv call costs ~1.5 ms whatever its size: a few hundred small chunks are gathered first

    def __init__(self, path):
        self.final, self.path, self.parts, self.held, self.n_held = path, path + '.partial', 0, [], 0

    def __call__(self, df, codes=None):
        if self._write_native(df, codes):
            return
        self.held.append(df)
        self.n_held += len(df)
        if self.n_held >= self.FLUSH_ROWS:
            self._flush()

    def _write_native(self, df, codes):
        """a chunk whose string columns came with dictionary codes (lgmi.mutual_information.regions_pair_mi_table) is
        formatted by liblgmi_io (lgio_write_table: the bytes to_csv writes, floats included — tested against numpy on
        millions of values); anything else, and a name pandas would quote, goes through pandas"""
        if not codes or os.environ.get('LGMI_PANDAS_TABLES'):
            return False
        columns = []
        for c in df.columns:
            if c in codes:
                columns.append((c,) + tuple(codes[c]))
            elif df[c].dtype.kind in 'iuf':
                columns.append((c, df[c].to_numpy()))
            else:
                return False
        self._flush()                                       # (parts are appended in the order they came)
        from .io import write_table
        try:
            write_table(self.path, columns, header=(self.parts == 0), append=(self.parts > 0))
        except ValueError:
            return False
        self.parts += 1
        return True

    def _flush(self):
        if self.held:
            df = self.held[0] if len(self.held) == 1 else pd.concat(self.held, ignore_index=True)
            df.to_csv(self.path, sep='\t', index=False, header=(self.parts == 0), mode='a' if self.parts else 'w')
            self.parts, self.held, self.n_held = self.parts + 1, [], 0

    def close(self, whole):
        """whole: the run's complete pair table — written here when no chunk came through the sink (serial runs, empty tables)"""
        self._flush()
        if not self.parts:
            whole.to_csv(self.path, sep='\t', index=False)
        os.replace(self.path, self.final)

    def abort(self):
        if os.path.exists(self.path):
            os.remove(self.path)


def _write_removed(df, path, header=True, append=False):
    """the removed-site table (strings and one integer column, millions of rows: one per covered position of every
    footprint) as pandas' to_csv(sep='\t', index=False) writes it, byte for byte, through pyarrow's CSV writer when it is
    there: the strings go in as dictionary codes (3.3 s -> 1.1 s for 3.85 M rows)"""
    try:
        import pyarrow as pa
        import pyarrow.csv as pc
    except ImportError:
        pa = None
    import numpy as np
    import pandas as pd
    cols = list(df.columns)
    mode = 'a' if append else 'w'
    if pa is None or cols != ['chromosome', 'strand', 'pos', 'removed'] or df['pos'].dtype != np.int64 or len(df) == 0:
        df.to_csv(path, sep='\t', index=False, header=header, mode=mode)
        return
    arrays = []
    for c in cols:
        if c == 'pos':
            arrays.append(pa.array(df[c].to_numpy()))
            continue
        if isinstance(df[c].dtype, pd.CategoricalDtype):
            codes, uniq = df[c].cat.codes.to_numpy(), list(df[c].cat.categories)
            if (codes < 0).any():                                          # a missing value: pandas writes an empty field
                df.to_csv(path, sep='\t', index=False, header=header, mode=mode)
                return
        else:
            codes, uniq = pd.factorize(df[c].to_numpy())
        if any(not isinstance(u, str) or any(ch in u for ch in '\t\n\r"') for u in uniq):
            df.to_csv(path, sep='\t', index=False, header=header, mode=mode)     # something pandas would quote
            return
        arrays.append(pa.DictionaryArray.from_arrays(pa.array(codes.astype(np.int32)), pa.array(list(uniq))).cast(pa.string()))
    with open(path, 'ab' if append else 'wb') as f:
        if header:
            f.write(('\t'.join(cols) + '\n').encode())
        pc.write_csv(pa.Table.from_arrays(arrays, names=cols), f,
                     pc.WriteOptions(delimiter='\t', quoting_style='none', include_header=False))


def get_footprints(sam, chromosomes, min_read_count=2):
    """[chromosome, start, end, read_count] of merged read intervals with enough reads
    (src/giremi/footprint.py:6-50: sort by start, fuse while the next start <= the running end).
    With the indexed reader the intervals come as two numpy arrays straight from the BAM records and the merge
    is a running maximum — no per-read Python object."""
    import numpy as np
    out = []
    for chrom in chromosomes:
        try:
            if hasattr(sam, 'intervals'):
                starts, ends = sam.intervals(chrom)
            else:
                spans = [(r.reference_start, r.reference_end) for r in sam.fetch(chrom)]
                starts = np.array([a for a, _b in spans], np.int64)
                ends = np.array([b for _a, b in spans], np.int64)
        except (ValueError, KeyError):           # contig absent from the BAM
            continue
        if len(starts) == 0:
            continue
        if len(starts) > 1 and bool((starts[1:] < starts[:-1]).any()):
            order = np.argsort(starts, kind='stable')
            s, e = starts[order], ends[order]
        else:                                        # a coordinate-sorted file hands them over in order (4 M reads: the sort
            s, e = starts, ends                      # and the two gathers were 0.3 s of the scan's 1.1)
        reach = np.maximum.accumulate(e)
        first = np.concatenate([[True], s[1:] > reach[:-1]])           # a read starting beyond everything before it
        idx = np.nonzero(first)[0]
        last = np.concatenate([idx[1:], [len(s)]]) - 1
        for lo, hi, n in zip(s[idx].tolist(), reach[last].tolist(), (last - idx + 1).tolist()):
            if n >= min_read_count:
                out.append([chrom, lo, hi, n])
    return out


_WORKER_HANDLES = {}          # (bam, fasta) -> (sam, genome) of THIS process


class _Reopen:
    """picklable: a pool worker opens its own handles on the alignment and genome files — once per worker process, not
    once per job (a job is a few dozen footprints; every open read the .bai again, and a FASTA without a .fai was indexed
    by a pass over the whole file in every job: a fifth of a second per job on a 12-MB genome).  ``fai``: the index the
    parent's FastaReader already holds."""

    def __init__(self, bam, fasta, fai=None):
        self.bam, self.fasta, self.fai = bam, fasta, fai

    def __call__(self):
        key = (os.getpid(), self.bam, self.fasta)
        got = _WORKER_HANDLES.get(key)
        if got is None:
            _WORKER_HANDLES.clear()                            # (a forked child inherits the parent's entry under the parent's pid)
            got = _WORKER_HANDLES[key] = (open_alignment(self.bam), open_fasta(self.fasta, fai=self.fai))
        return got


def read_repeats(path):
    """chrom -> [[start, end]] sorted by start (src/giremi/fileio.py:4-21)"""
    table = defaultdict(list)
    if path:
        with open(path) as f:
            for line in f:
                if line.startswith('#'):
                    continue
                c = line.split('\t')
                table[c[0]].append([int(c[1]), int(c[2])])
        for v in table.values():
            v.sort(key=lambda iv: iv[0])
    return table


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(format='%(asctime)s %(levelname)s %(message)s', level=logging.INFO)
    if not args.mi_calculation_only and not args.mip_table:
        sys.exit('lgmi covers the MI step: run with --mi_calculation_only, or with --mip_table for the mismatch table with mean_mi '
                 'and mip that the reference builds before its GLM (GLM scoring, stat.py:32-143, is not part of lgmi)')
    if not args.skip_strand_correction:
        sys.exit('GTF-based strand correction (strand.py) is not part of lgmi: run with --skip_strand_correction')
    if not args.genome_fasta:
        sys.exit('--genome_fasta is required')

    import time
    timing = {}
    t_all = t0 = time.perf_counter()
    import os
    from . import region as _region                      # (loaded before the workers are forked: they run its _extract_chunk)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if args.gpus is not None and args.gpus != world:
        sys.exit('--gpus %d: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node %d -m lgmi.cli ...); '
                 'WORLD_SIZE is %d' % (args.gpus, args.gpus, world))
    group = None
    if world > 1:
        from .dist import group_from_env
        group = group_from_env()                         # plain sockets: the RCCL id, barriers, small gathers
    # the extraction workers of a one-rank run are forked HERE, while this process is still small (a fork copies the page
    # tables of everything loaded so far: sixteen of them after the index, the variants and 8,000 footprints were a quarter
    # of a second) and long before the HIP context exists; they wait for their jobs (regions_mismatch_analysis: pool)
    pool = None
    if world == 1 and args.thread and args.thread > 1:
        import multiprocessing as mp
        pool = mp.get_context('fork').Pool(args.thread)
    try:
        return _run(args, timing, t_all, t0, world, rank, group, pool)
    finally:
        if pool is not None:
            pool.terminate()
            pool.join()


def _run(args, timing, t_all, t0, world, rank, group, pool):
    import os
    import time
    from .engine import Engine
    from .region import regions_mismatch_analysis
    sam = open_alignment(args.bam_file)
    # the footprint scan (native, on its own threads, no interpreter lock) runs while this thread indexes the genome and
    # reads the variants
    import threading
    logging.info('Get regions that are covered by enough reads.')
    scan = {}

    def scan_footprints():
        try:
            scan['footprints'] = get_footprints(sam, args.chromosomes, args.min_total_depth)
        except BaseException as e:                          # noqa: BLE001 — raised again below, in the thread that owns the run
            scan['error'] = e
    scanner = threading.Thread(target=scan_footprints, name='lgmi-footprints')
    scanner.start()
    try:
        genome = open_fasta(args.genome_fasta)
        vcf = open_variants(args.snp_bcf) if args.snp_bcf else None
        repeats = read_repeats(args.repeat_txt)
    finally:
        scanner.join()
    if 'error' in scan:
        raise scan['error']
    footprints = scan['footprints']
    if world > 1:
        # contiguous runs of footprints balanced by read count (the reference cuts the footprint list into contiguous
        # chunks too, script/giremi.py:367-370): rank order stays footprint order, which is the output order
        cum = np.cumsum([f[3] for f in footprints]) if footprints else np.zeros(0)
        total = float(cum[-1]) if len(cum) else 0.0
        cuts = [int(np.searchsorted(cum, total * r / world, side='left')) for r in range(world)] + [len(footprints)]
        footprints = footprints[cuts[rank]:cuts[rank + 1]]
    timing['open_and_footprints_s'] = time.perf_counter() - t0
    timing['n_footprints'] = len(footprints)
    t0 = time.perf_counter()
    logging.info('Calculate mismatches in each region.')
    jobs = []
    for chrom, start, end, _n in footprints:
        snps = sorted(r.start for r in vcf.fetch(chrom, start, end)) if vcf is not None else []
        # The reference hands every footprint the repeat intervals that lie OUTSIDE it (script/giremi.py:55-59: a > end or
        # b < start).  A site of the footprint comes from a read that overlaps it, and the footprint is the union of the
        # reads that overlap each other: start <= pos < end.  An interval that begins after `end` or ends before `start`
        # cannot contain such a position, so the filter it feeds (mismatch.py:314-323) never fires — reproduced as the
        # empty list, not as a list of (all repeats of the chromosome) per footprint, which is quadratic over a run and
        # would be pickled to every worker.
        reps = []
        jobs.append({'chromosome': chrom, 'start': start, 'end': end, 'snp_positions': snps,
                     'simple_repeat_intervals': reps, 'read_strand_dict': None})
    timing['footprint_inputs_s'] = time.perf_counter() - t0
    made = []

    def make_engine():                      # after the worker pool is gone: a HIP context does not survive fork()
        device = args.device
        if device is None and world > 1:
            import ctypes
            from . import _lib
            n_dev = ctypes.c_int(0)
            device = int(os.environ.get('LOCAL_RANK', '0'))
            if _lib.load().lgmi_device_count(ctypes.byref(n_dev)) == 0 and n_dev.value > 0:
                device %= n_dev.value
        made.append(Engine(device))
        if group is not None:
            made[0].comm_init_group(group)
        return made[0]
    # one rank: the removed-site table (one row per covered position: most of what a run writes) goes out part by part
    # while later footprints are still being extracted
    fai = getattr(genome, 'index', None)
    if fai is not None and len(fai) > 4096:
        fai = None                                         # (it travels with every job: a transcriptome's 200,000 entries would not pay; the workers index once each)
    removed_writer = _RemovedWriter(args.output_prefix + '.removed.txt') if world == 1 else None
    pairs_writer = _PairsWriter(args.output_prefix + '.mi.txt') if world == 1 else None
    try:
        df_sites, df_mi, df_removed = regions_mismatch_analysis(
            jobs, sam, genome, min_common_reads=args.mi_min_common_read, n_shuffles=args.n_shuffles, seed=args.seed,
            engine=make_engine, concat=True, threads=args.thread, reopen=_Reopen(args.bam_file, args.genome_fasta, fai=fai), timing=timing,
            group=group, removed_sink=removed_writer, pairs_sink=pairs_writer, pool=pool,
            keep_non_spliced_read=args.keep_non_spliced_read,
            min_dist_from_splice=args.min_dist_from_splice, min_allele_depth=args.min_allele_depth,
            min_allele_ratio=args.min_allele_ratio, min_total_depth=args.min_total_depth,
            homopoly_length=args.homopoly_length, min_het_snp_ratio=args.min_het_snp_ratio,
            max_het_snp_ratio=args.max_het_snp_ratio, mismatch_window_size=args.mismatch_window_size,
            max_window_mismatch=args.max_window_mismatch, max_window_mismatch_type=args.max_window_mismatch_type,
            mode=args.mode)
    except BaseException:
        if removed_writer is not None:
            removed_writer.abort()                         # no half-written table next to missing .mi.txt / .strand.txt files
        if pairs_writer is not None:
            pairs_writer.abort()
        raise
    t0 = time.perf_counter()
    strand_df = pd.DataFrame.from_records([], columns=['read_name', 'original_read_strand', 'corrected_read_strand'])
    if world > 1:
        # every rank writes its part of the removed-site table next to the output; rank 0 stitches the parts in rank
        # order (= footprint order) once all are there, and writes the gathered pair table
        part = '%s.removed.txt.rank%d' % (args.output_prefix, rank)
        _write_removed(df_removed, part, header=(rank == 0))
        group.barrier()
        if rank == 0:
            import shutil
            with open(args.output_prefix + '.removed.txt', 'wb') as out:
                for r in range(world):
                    with open('%s.removed.txt.rank%d' % (args.output_prefix, r), 'rb') as f:
                        shutil.copyfileobj(f, out)
            for r in range(world):
                os.remove('%s.removed.txt.rank%d' % (args.output_prefix, r))
            strand_df.to_csv(args.output_prefix + '.strand.txt', sep='\t', index=False)
            df_mi.to_csv(args.output_prefix + '.mi.txt', sep='\t', index=False)
        group.barrier()
    else:
        strand_df.to_csv(args.output_prefix + '.strand.txt', sep='\t', index=False)
        pairs_writer.close(df_mi)                        # (its parts were written as the run went; a serial run's table here)
        removed_writer.close()
    if args.mip_table:
        # script/giremi.py:415-429: mip = ECDF of the het-SNP rows' mean_mi (those that have one) at every row's mean_mi.
        # Several ranks: the ECDF needs every rank's het-SNP means — the site tables go to rank 0 through the socket group.
        from .stat import mean_mi_to_mip
        table = df_sites
        if world > 1:
            parts = group.gather({c: (df_sites[c].to_numpy(np.float64) if df_sites[c].dtype.kind in 'fiu' else df_sites[c].astype(str).tolist())
                                  for c in df_sites.columns})
            if rank == 0:
                table = pd.concat([pd.DataFrame(p_, columns=df_sites.columns) for p_ in parts], axis=0, ignore_index=True)
                for c in ('pos', 'depth'):
                    table[c] = table[c].astype(np.int64)
        if rank == 0:
            table = table.copy()
            eng = made[0] if made else None
            table['mip'] = mean_mi_to_mip(table['mean_mi'].to_numpy(np.float64), table['type'].to_numpy(), engine=eng) \
                if len(table) else np.zeros(0)
            table.to_csv(args.output_prefix + '.mismatch_mip.txt', sep='\t', index=False)
    timing['write_s'] = time.perf_counter() - t0
    for e in made:
        e.close()
    timing['total_s'] = time.perf_counter() - t_all
    timing['mi_rows'], timing['site_rows'] = (len(df_mi) if df_mi is not None else 0), len(df_sites)
    timing['rank'], timing['world'] = rank, world
    if group is not None:
        group.close()
    if args.timing_json and rank == 0:
        import json
        with open(args.timing_json, 'w') as f:
            json.dump({k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()}, f)
    logging.info('All done!')
    return df_sites, df_mi, df_removed


if __name__ == '__main__':
    main()
