"""The caller of the hot path: per-footprint site extraction + filters + MI block.

Drop-in for ``get_region_mismatches_with_filters`` and ``region_mismatch_analysis``
of the reference (src/giremi/mismatch.py:11-342 and :345-509), written from scratch.
Same signatures, same three DataFrames (mismatch table, pair-MI table, removed table),
same quirks (listed where they are reproduced); the MI block (:384-404) runs on the
MI355X through ``region_pair_mi`` instead of the per-pair Python/sklearn loop.

`sam` and `genome` are duck-typed like pysam's AlignmentFile / FastaFile:
``sam.fetch(chrom, start, end)`` -> reads with ``query_name``, ``reference_start``,
``is_reverse``, ``get_tag('cs')``; ``sam.pileup(contig=, start=, stop=)`` -> columns with
``pos``, ``get_query_names()``, ``get_query_sequences()``; ``genome.fetch(chrom, a, b)``.
Only ``mode='cs'`` (minimap2 ``--cs`` tags) is supported here; the CIGAR+MD conversion
of the reference (src/giremi/cs.py:110-363) is not part of this path.
"""
from __future__ import annotations

import os
import re
from bisect import bisect_left
from collections import defaultdict
from typing import List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from .mutual_information import region_pair_mi, regions_pair_mi, regions_pair_mi_table, regions_pair_mi_table_dist

_CS_TOKEN = re.compile(r'([:*+\-~])([0-9a-z]+)')
_COMPLEMENT = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'N': 'N'}
_COMP4 = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}    # the filter stage knows no 'N' (KeyError, like mismatch.py:151)


def cs_operations(cs_string: str) -> List[Tuple[int, int, str, str]]:
    """minimap2 short cs tag -> [(low, high, op, value)] with reference offsets relative to the
    alignment start (src/giremi/cs.py:8-41): ':' n matches, '*xy' one substitution x->y, '+seq'
    insertion (no reference span), '-seq' deletion, '~aaNNbb' intron of NN bases."""
    out, at = [], 0
    for op, value in _CS_TOKEN.findall(cs_string):
        if op == ':':
            span = int(value)
        elif op == '*':
            span = 1
        elif op == '+':
            span = 0
        elif op == '-':
            span = len(value)
        else:
            span = int(''.join(ch for ch in value if ch.isdigit()))
        out.append((at, at + span, op, value))
        at += span
    return out


def _merge(intervals):
    """src/giremi/utils.py:4-16 — sort by start, fuse when the next start is <= the running end"""
    merged: List[List[int]] = []
    for lo, hi in sorted(intervals, key=lambda iv: iv[0]):
        if merged and lo <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], hi)
        else:
            merged.append([lo, hi])
    return merged


def _inside(positions, intervals) -> List[bool]:
    """src/giremi/utils.py:19-29 — start <= pos < end of some interval, by the same two
    right-bisections (a position counts when #starts<=pos exceeds #ends<=pos by exactly one);
    like the reference it sorts the caller's interval list in place"""
    if len(positions) == 0:
        return []
    intervals.sort(key=lambda iv: iv[0])
    starts = [iv[0] for iv in intervals]
    ends = [iv[1] for iv in intervals]
    a = np.searchsorted(starts, positions, side='right')
    b = np.searchsorted(ends, positions, side='right')
    return [int(x) - int(y) == 1 for x, y in zip(a, b)]


def _new_site(with_removed=False):
    site = {'ref': '', 'type': 'mismatch', 'depth': defaultdict(int), 'nt': defaultdict(list),
            'neighbor': defaultdict(int), 'up': '', 'down': ''}
    if with_removed:
        site['removed'] = ''
    return site


def _context_steps(sites, gone, chromosome, genome, homopoly_length, simple_repeat_intervals, snp_positions,
                   min_het_snp_ratio, max_het_snp_ratio):
    """steps 7-9 of the site extraction on the sites of one strand that survived the BAM-only steps: they need the
    genome and the caller's lists, and see only a handful of sites (mismatch.py:292-340)"""
    # ---- 7. homopolymer context; flanking bases (:292-312)
    k = int(homopoly_length / 2)
    for pos in sorted(sites.keys()):
        left = genome.fetch(chromosome, pos - homopoly_length, pos).upper()
        right = genome.fetch(chromosome, pos + 1, pos + homopoly_length + 1).upper()
        sites[pos]['up'] = left[-1].upper()
        sites[pos]['down'] = right[0].upper()
        if len(set(left)) == 1 or len(set(right)) == 1 or len(set(left[-k:] + right[0:k])) == 1:
            gone[pos] = sites.pop(pos)
            gone[pos]['removed'] = 'in homopoly regions'
    # ---- 8. simple repeats (:314-323)
    here = sorted(sites.keys())
    for pos, hit in zip(here, _inside(here, simple_repeat_intervals)):
        if hit:
            gone[pos] = sites.pop(pos)
            gone[pos]['removed'] = 'in simple repeat regions'
    # ---- 9. depth = surviving alleles only; SNP typing (:325-340)
    for pos in sorted(sites.keys()):
        for nt in list(sites[pos]['depth'].keys()):
            sites[pos]['depth'].pop(nt)
        for nt in list(sites[pos]['nt'].keys()):
            sites[pos]['depth'][nt] = len(sites[pos]['nt'][nt])
    for pos in sorted(sites.keys()):
        if pos in snp_positions:
            total = sum(sites[pos]['depth'].values())
            top = max(d / total for d in sites[pos]['depth'].values())
            sites[pos]['type'] = 'het_snp' if min_het_snp_ratio <= top <= max_het_snp_ratio else 'snp'


def get_region_mismatches_with_filters(chromosome, start_pos, end_pos, sam, genome,
                                       keep_non_spliced_read=False, min_dist_from_splice=4,
                                       min_allele_depth=3, min_allele_ratio=0.1, min_total_depth=6,
                                       homopoly_length=5, simple_repeat_intervals=[], snp_positions=[],
                                       read_strand_dict=None, min_het_snp_ratio=0.35, max_het_snp_ratio=0.65,
                                       mismatch_window_size=100, max_window_mismatch=10,
                                       max_window_mismatch_type=3, mode='cs'):
    """-> (mismatches, removed_mismatches), each {'+': {pos: site}, '-': {pos: site}} (mismatch.py:11-342)."""
    if mode != 'cs':
        raise NotImplementedError("only mode='cs' is supported by lgmi.region")
    # auto-creating maps, like the reference: merely LOOKING a position up creates an empty site, which
    # the depth filter later reports as removed — reproduced on purpose (mismatch.py:29-62, :166)
    kept = {s: defaultdict(_new_site) for s in '+-'}
    dropped = {s: defaultdict(lambda: _new_site(True)) for s in '+-'}
    if read_strand_dict is None:
        read_strand_dict = {}

    # ---- 1. substitutions of every (spliced) read, minus those next to a splice junction (:69-149)
    for read in sam.fetch(chromosome, start_pos, end_pos):
        strand = '-' if read.is_reverse else '+'
        if read.query_name in read_strand_dict:
            strand = read_strand_dict[read.query_name]
        else:
            read_strand_dict[read.query_name] = strand
        ops = cs_operations(read.get_tag('cs'))
        origin = read.reference_start
        subs = sorted(([lo + origin, val] for lo, _hi, op, val in ops if op == '*'), key=lambda t: t[0])
        introns = sorted(([lo + origin, hi + origin] for lo, hi, op, _v in ops if op == '~'), key=lambda t: t[0])
        if not keep_non_spliced_read and not introns:
            continue
        if not subs:
            continue
        if introns and min_dist_from_splice > 0:
            junctions = sorted([iv[0] for iv in introns] + [iv[1] for iv in introns])
            near = _merge([[j - min_dist_from_splice, j + min_dist_from_splice] for j in junctions])
            flags = _inside([t[0] for t in subs], near)
        else:
            flags = [False] * len(subs)
        for (pos, refalt), close in zip(subs, flags):
            if close:
                continue
            kept[strand][pos]['ref'] = refalt[0].upper()
            kept[strand][pos]['nt'][refalt[1].upper()].append(read.query_name)

    for strand in '+-':
        sites = kept[strand]
        gone = dropped[strand]
        if len(sites) == 0:
            continue
        # ---- 2. reads that carry the reference base, from the pile-up (:160-190)
        known = set(sites.keys())                        # (the reference tests membership in a sorted LIST, :159-167:
        #                                                  the same answer in time linear in the number of sites per column)
        for column in sam.pileup(contig=chromosome, start=start_pos, stop=end_pos):
            pos = column.pos
            ref = sites[pos]['ref']                      # creates an empty site for a new position
            if pos in known and ref.upper() in _COMP4:
                names = column.get_query_names()
                bases = [b.upper() for b in column.get_query_sequences()]
                same_strand = [read_strand_dict[n] == strand for n in names]      # KeyError for an unknown read
                sites[pos]['nt'][ref].extend(n for n, b, ok in zip(names, bases, same_strand) if b == ref and ok)
        # ---- 3. allele depths (:203-208)
        for pos in sorted(sites.keys()):
            for nt in list(sites[pos]['nt'].keys()):
                sites[pos]['depth'][nt] = len(sites[pos]['nt'][nt])
        # ---- 4. too many substitutions of too many kinds in the window (:211-240)
        half = round(mismatch_window_size / 2)
        snapshot = sorted(sites.keys())
        # What a site adds to its neighbours' counts: one change string per non-reference allele of its depth dict.  Most
        # sites of a footprint are the empty ones the pile-up loop created (one per covered position) and add nothing:
        # only the contributing positions are visited (round 3: with every position of the window visited this loop was
        # 80 % of the site extraction on a 2,000-gene BAM).  Same counts, same insertion order, same side effects: a site
        # removed earlier in the loop contributes nothing and is re-created empty by the look-up (mismatch.py:214-222).
        contrib = {}
        for q in snapshot:
            ref = sites[q]['ref']
            alts = [a for a in sites[q]['depth'] if a != ref]
            if alts:
                contrib[q] = ['%s>%s' % (ref, nt) if strand == '+' else '%s>%s' % (_COMP4[ref], _COMP4[nt]) for nt in alts]
        cpos = sorted(contrib)
        removed_here, dead = [], set()                   # removed in this loop, in increasing position
        for pos in snapshot:
            lo, hi = pos - half, pos + half
            if bisect_left(snapshot, hi) - bisect_left(snapshot, lo) < 2:      # nobody but pos itself in the window
                continue
            for q in removed_here[bisect_left(removed_here, lo):bisect_left(removed_here, hi)]:
                sites[q]['ref']                          # a site removed earlier in this loop comes back empty
            neighbor = sites[pos]['neighbor']
            for q in cpos[bisect_left(cpos, lo):bisect_left(cpos, hi)]:
                if q == pos or q in dead:
                    continue
                for change in contrib[q]:
                    neighbor[change] += 1
            if sum(neighbor.values()) > max_window_mismatch and len(neighbor) > max_window_mismatch_type:
                gone[pos] = sites.pop(pos)
                gone[pos]['removed'] = 'too many window mismatches'
                removed_here.append(pos)
                dead.add(pos)
        # ---- 5. shallow alleles, rare alleles (the depth dict keeps every allele: :243-266)
        for pos in sorted(sites.keys()):
            for nt in list(sites[pos]['nt'].keys()):
                if sites[pos]['depth'][nt] < min_allele_depth:
                    sites[pos]['nt'].pop(nt)
        for pos in sorted(sites.keys()):
            total = sum(sites[pos]['depth'].values())
            for nt in list(sites[pos]['nt'].keys()):
                if sites[pos]['depth'][nt] / total < min_allele_ratio:
                    sites[pos]['nt'].pop(nt)
        # ---- 6. shallow sites, single-allele sites (:268-290)
        for pos in sorted(sites.keys()):
            if sum(sites[pos]['depth'].values()) < min_total_depth:
                gone[pos] = sites.pop(pos)
                gone[pos]['removed'] = 'too few usable reads after filters'
        for pos in sorted(sites.keys()):
            if len(sites[pos]['nt']) < 2:
                gone[pos] = sites.pop(pos)
                gone[pos]['removed'] = 'not enough allele after filters'
        _context_steps(sites, gone, chromosome, genome, homopoly_length, simple_repeat_intervals, snp_positions,
                       min_het_snp_ratio, max_het_snp_ratio)
    return kept, dropped


def region_mismatch_analysis(chromosome, start_pos, end_pos, sam, genome,
                             keep_non_spliced_read=False, min_dist_from_splice=4, min_allele_depth=3,
                             min_allele_ratio=0.1, min_total_depth=6, homopoly_length=5,
                             simple_repeat_intervals=[], snp_positions=[], read_strand_dict=None,
                             min_het_snp_ratio=0.35, max_het_snp_ratio=0.65, mismatch_window_size=100,
                             max_window_mismatch=10, max_window_mismatch_type=3, mode='cs',
                             min_common_reads=5, n_shuffles=0, seed=0, engine=None):
    """-> (df_mismatches, df_mismatch_pair_mi, df_removed_mismatches) as mismatch.py:345-509.
    With ``n_shuffles`` > 0 the pair table gains a ``p_perm`` column (permutation p-value, new)."""
    sites, gone = get_region_mismatches_with_filters(
        chromosome=chromosome, start_pos=start_pos, end_pos=end_pos, sam=sam, genome=genome,
        keep_non_spliced_read=keep_non_spliced_read, min_dist_from_splice=min_dist_from_splice,
        min_allele_depth=min_allele_depth, min_allele_ratio=min_allele_ratio, min_total_depth=min_total_depth,
        homopoly_length=homopoly_length, simple_repeat_intervals=simple_repeat_intervals,
        snp_positions=snp_positions, read_strand_dict=read_strand_dict, min_het_snp_ratio=min_het_snp_ratio,
        max_het_snp_ratio=max_het_snp_ratio, mismatch_window_size=mismatch_window_size,
        max_window_mismatch=max_window_mismatch, max_window_mismatch_type=max_window_mismatch_type, mode=mode)

    # the MI block (:384-404) — one batched GPU call for both strands
    records, mean_mi, pvals = region_pair_mi(sites, chromosome, min_common_reads, n_shuffles=n_shuffles, seed=seed,
                                             engine=engine)
    return _frames(chromosome, sites, gone, records, mean_mi, pvals)


_LATE_REASONS = ('in homopoly regions', 'in simple repeat regions')
_ACGT = 'ACGT'


def _is_whole(x):
    return isinstance(x, (int, np.integer)) or (isinstance(x, float) and x.is_integer())


def region_sites_native(chromosome, start_pos, end_pos, sam, genome,
                        keep_non_spliced_read=False, min_dist_from_splice=4,
                        min_allele_depth=3, min_allele_ratio=0.1, min_total_depth=6,
                        homopoly_length=5, simple_repeat_intervals=[], snp_positions=[],
                        read_strand_dict=None, min_het_snp_ratio=0.35, max_het_snp_ratio=0.65,
                        mismatch_window_size=100, max_window_mismatch=10,
                        max_window_mismatch_type=3, mode='cs', read_ids=False):
    """``get_region_mismatches_with_filters`` with the BAM-only steps (1-6) in liblgmi_io (lgio_bam_region_sites) and
    the removed sites as arrays: -> (sites, (removed, reasons)) with ``sites`` what the Python routine returns as its
    first value (plain dicts per strand) and ``(removed, reasons)`` what ``_compact_gone`` makes of its second — or None
    when this footprint has to go through the Python routine (a caller-provided read_strand_dict, an alignment object
    without the native entry point, a cs string or a base the native walk does not cover).  tests/test_region_fast.py
    holds the two against each other.  ``read_ids``: the allele read lists hold integer read ids (the index of the first
    record of the read's NAME in the footprint, lgio_sites.read_uid) instead of name strings — what a caller that only
    packs the footprint needs, without ~500 Python strings per site made and hashed (0.28 of 1.9 s per 200 footprints)."""
    from .io import SiteParams, REMOVED_REASONS
    if (mode != 'cs' or read_strand_dict is not None or not hasattr(sam, 'region_sites')
            or not _is_whole(min_dist_from_splice) or start_pos is None or end_pos is None or start_pos < 0
            or end_pos < start_pos):
        return None
    params = SiteParams(keep_non_spliced_read=1 if keep_non_spliced_read else 0, min_base_quality=13, max_depth=8000,
                        min_dist_from_splice=int(min_dist_from_splice), half_window=int(round(mismatch_window_size / 2)),
                        min_allele_depth=min_allele_depth, min_allele_ratio=min_allele_ratio,
                        min_total_depth=min_total_depth, max_window_mismatch=max_window_mismatch,
                        max_window_mismatch_type=max_window_mismatch_type)
    raw = sam.region_sites(chromosome, start_pos, end_pos, params)
    if raw is None:
        return None
    noff, pool = raw['name_off'], raw['names']
    names = {}

    def name(i):
        got = names.get(i)
        if got is None:
            got = names[i] = pool[noff[i]:noff[i + 1]].decode()
        return got

    sites = {'+': {}, '-': {}}
    aoff, roff = raw['allele_off'], raw['reads_off']
    reads = (raw['read_uid'][raw['reads']] if read_ids else raw['reads']).tolist()
    ref_b, nt_b = raw['ref'].decode(), raw['allele_nt'].decode()
    for k in range(len(raw['pos'])):
        strand = '-' if raw['strand'][k] else '+'
        site = _new_site()
        site['ref'] = ref_b[k]
        for a in range(int(aoff[k]), int(aoff[k + 1])):
            ids = reads[int(roff[a]):int(roff[a + 1])]
            site['nt'][nt_b[a]] = ids if read_ids else [name(i) for i in ids]
        for t in np.flatnonzero(raw['neighbor'][k]).tolist():
            r, alt = _ACGT[t >> 2], _ACGT[t & 3]
            change = '%s>%s' % (r, alt) if strand == '+' else '%s>%s' % (_COMP4[r], _COMP4[alt])
            site['neighbor'][change] = int(raw['neighbor'][k][t])
        sites[strand][int(raw['pos'][k])] = site
    removed = {}
    reasons = list(REMOVED_REASONS) + list(_LATE_REASONS)
    for s, strand in enumerate('+-'):
        late = {}
        if sites[strand]:
            _context_steps(sites[strand], late, chromosome, genome, homopoly_length, simple_repeat_intervals,
                           snp_positions, min_het_snp_ratio, max_het_snp_ratio)
        pos, codes = raw['removed'][s]
        if late:
            pos = np.concatenate([pos, np.fromiter(late.keys(), np.int64, len(late))])
            codes = np.concatenate([codes, np.array([reasons.index(v['removed']) for v in late.values()], np.uint8)])
        removed[strand] = (pos, codes)
    return sites, (removed, reasons)


def _compact_gone(gone):
    """the removed-site table of one footprint as arrays: {strand: (positions int64, reason codes uint8)} + the reason
    strings — a footprint auto-creates an (empty, later "removed") site for every covered position (mismatch.py:29-62,
    :166), so a whole run's removed table is millions of rows: they must not travel, or be assembled, as Python dicts"""
    reasons, code = [], {}
    out = {}
    for strand in '+-':
        d = gone[strand]
        pos = np.fromiter(d.keys(), np.int64, len(d))
        codes = np.empty(len(d), np.uint8)
        for k, site in enumerate(d.values()):
            r = site['removed']
            c = code.get(r)
            if c is None:
                c = code[r] = len(reasons)
                reasons.append(r)
            codes[k] = c
        out[strand] = (pos, codes)
    return out, reasons


_STRIP_READS = True      # (tests that stand a CPU oracle in for the MI step need the read lists back from the workers)


def _extract_chunk(job, sam=None, genome=None):
    """site extraction + filters for a list of footprints -> [(chromosome, sites, gone)] (pool worker and serial path);
    with job[3] (compact) `gone` is the array form of _compact_gone"""
    reopen, footprints, filter_kwargs = job[:3]
    compact = len(job) > 3 and job[3]
    pack = len(job) > 4 and job[4]
    import time
    t_job = time.time()
    if reopen is not None:
        sam, genome = reopen()
    out = []
    for fp in footprints:
        args = dict(chromosome=fp['chromosome'], start_pos=fp['start'], end_pos=fp['end'], sam=sam, genome=genome,
                    snp_positions=fp.get('snp_positions', []), simple_repeat_intervals=fp.get('simple_repeat_intervals', []),
                    read_strand_dict=fp.get('read_strand_dict'), **filter_kwargs)
        # a whole run (compact): the BAM-only steps natively, when the footprint is one the native walk covers
        fast = region_sites_native(read_ids=bool(pack and _STRIP_READS), **args) if compact and not os.environ.get('LGMI_PY_SITES') else None
        if fast is not None:
            out.append((fp['chromosome'],) + fast)
            continue
        sites, gone = get_region_mismatches_with_filters(**args)
        if compact:
            gone = _compact_gone(gone)
        if reopen is not None:              # results cross a process boundary: plain dicts (the site factory does not pickle)
            sites = {strand: dict(d) for strand, d in sites.items()}
            if not compact:
                gone = {strand: dict(d) for strand, d in gone.items()}
        out.append((fp['chromosome'], sites, gone))
    if pack:
        # a worker of a whole run packs its own footprints (the parent concatenates the chunks' batches): the packing is
        # spread over the pool like the extraction, and the read-name lists — most of what a chunk's result weighs, and of
        # no use after the packing — stay here
        from .pack import pack_blocks
        blocks = []
        for _chrom, sites, _gone in out:
            blocks.extend(sites.get(s, {}) for s in '+-')
        batch = pack_blocks(blocks) if blocks else None
        if _STRIP_READS:
            for _chrom, sites, _gone in out:
                for strand in '+-':
                    for site in sites[strand].values():
                        site['nt'] = {a: len(v) for a, v in site['nt'].items()}     # (allele order kept; _site_rows reads depth)
        # ... and builds what the parent would otherwise build footprint by footprint while the pool waits for it (round 5:
        # with the extraction itself three times faster the parent's 0.8 s of site rows and 0.7 s of removed-site arrays per
        # 8,000 footprints were the pipeline's longest leg): the rows of the mismatch table, their mean_mi still empty (the
        # parent fills the column in from the GPU's per-site means), and the chunk's removed sites as four flat arrays
        nothing = {'+': {}, '-': {}}
        extras = {'site_rows': [_site_rows(chrom, sites, nothing) for chrom, sites, _gone in out],
                  'removed': _removed_arrays(out) if compact else None}
        if os.environ.get('LGMI_TRACE_JOBS'):                   # when this job ran, and where (the pipeline's parent prints them)
            extras['job'] = (os.getpid(), t_job, time.time())
        if len(job) > 5 and job[5] and compact:
            # the pipeline's parent reads the removed sites from extras['removed'] only: the per-footprint arrays — the same
            # 15 million rows once more — need not cross the process boundary (they were half of what a job sent back)
            out = [(chrom, sites, None) for chrom, sites, _gone in out]
        return out, batch, extras
    return out


def _removed_arrays(staged):
    """the removed sites of a chunk of footprints as flat arrays with chunk-local codes:
    -> (chromosome names, reason strings, chromosome code int32[n], strand int8[n], pos int64[n], reason code int8[n])"""
    chroms, reasons = {}, {}
    g_chrom, g_strand, g_pos, g_reason = [], [], [], []
    for chrom, _sites, (gone, rs) in staged:
        cc = chroms.setdefault(chrom, len(chroms))
        remap = np.array([reasons.setdefault(r, len(reasons)) for r in rs], np.int8)
        for k, strand in enumerate('+-'):
            pos, codes = gone[strand]
            if len(pos):
                g_chrom.append(np.full(len(pos), cc, np.int32))
                g_strand.append(np.full(len(pos), k, np.int8))
                g_pos.append(pos)
                g_reason.append(remap[codes])
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
    return list(chroms), list(reasons), cat(g_chrom, np.int32), cat(g_strand, np.int8), cat(g_pos, np.int64), cat(g_reason, np.int8)


def _removed_frame_from_arrays(arrays, chrom_code, reason_code):
    """_removed_frame of a chunk whose worker already flattened it (_removed_arrays): the chunk's codes mapped onto the run's"""
    chroms, reasons, g_chrom, g_strand, g_pos, g_reason = arrays
    cmap = np.array([chrom_code.setdefault(c, len(chrom_code)) for c in chroms] or [0], np.int32)
    rmap = np.array([reason_code.setdefault(r, len(reason_code)) for r in reasons] or [0], np.int8)
    as_cat = lambda codes, names: pd.Categorical.from_codes(codes, categories=list(names))
    return pd.DataFrame({'chromosome': as_cat(cmap[g_chrom], chrom_code), 'strand': as_cat(g_strand, '+-'),
                         'pos': g_pos, 'removed': as_cat(rmap[g_reason], reason_code)},
                        columns=['chromosome', 'strand', 'pos', 'removed'])


def _removed_frame(staged, chrom_code, reason_code):
    """the removed-site table of some footprints: one row per covered position — tens of millions in a run — with its three
    string columns as categoricals from small-integer codes (values, order and what to_csv writes are those of plain
    string columns)"""
    g_chrom, g_strand, g_pos, g_reason = [], [], [], []
    for chrom, _sites, (gone, reasons) in staged:
        cc = chrom_code.setdefault(chrom, len(chrom_code))
        remap = np.array([reason_code.setdefault(r, len(reason_code)) for r in reasons], np.int8)
        for k, strand in enumerate('+-'):
            pos, codes = gone[strand]
            if len(pos):
                g_chrom.append(np.full(len(pos), cc, np.int32))
                g_strand.append(np.full(len(pos), k, np.int8))
                g_pos.append(pos)
                g_reason.append(remap[codes])
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)
    as_cat = lambda parts, dt, names: pd.Categorical.from_codes(cat(parts, dt), categories=list(names))
    return pd.DataFrame({'chromosome': as_cat(g_chrom, np.int32, chrom_code), 'strand': as_cat(g_strand, np.int8, '+-'),
                         'pos': cat(g_pos, np.int64), 'removed': as_cat(g_reason, np.int8, reason_code)},
                        columns=['chromosome', 'strand', 'pos', 'removed'])


def _ordered_parts(pool, fn, jobs, poll_s=0.5):
    """``pool.imap(fn, jobs)`` that notices a worker's death.  multiprocessing.Pool replaces a dead worker by forking the
    parent again — here a parent that holds a HIP context and its runtime threads by then (advice r4) — and the dead
    worker's job never comes back, so a plain imap would wait for ever.  The set of worker pids is taken when the pool is
    new and looked at between results: any change ends the run with an error instead."""
    import multiprocessing as mp
    if not hasattr(pool, '_pool'):                              # (not the multiprocessing.Pool this was written against)
        yield from pool.imap(fn, jobs)
        return
    pids = sorted(p.pid for p in list(pool._pool))
    it = pool.imap(fn, jobs)
    while True:
        try:
            part = it.next(timeout=poll_s)
        except StopIteration:
            return
        except mp.TimeoutError:
            part = None
        now = sorted(p.pid for p in list(pool._pool) if p.exitcode is None)
        if now != pids:
            pool.terminate()
            raise RuntimeError('an extraction worker died (worker pids %s, now %s: out of memory?); the run was stopped — '
                               'a replacement would have been forked from a process that holds a GPU context' % (pids, now))
        if part is not None:
            yield part


def regions_mismatch_analysis(footprints, sam, genome, min_common_reads=5, n_shuffles=0, seed=0, engine=None,
                              concat=False, threads=1, reopen=None, timing=None, group=None, removed_sink=None,
                              pairs_sink=None, pool=None, **filter_kwargs):
    """``region_mismatch_analysis`` over many footprints with the MI blocks of many footprints per GPU batch — the shape the
    reference's per-chunk loop (src/giremi/script/giremi.py:32-88) takes when the MI step is a device call.

    ``footprints``: iterable of dicts with ``chromosome``, ``start``, ``end`` and optionally the per-footprint inputs
    ``snp_positions``, ``simple_repeat_intervals``, ``read_strand_dict``; ``filter_kwargs`` are the common filter
    parameters of ``region_mismatch_analysis``.  Returns a list of (df_sites, df_pairs, df_removed) in footprint order,
    or with ``concat=True`` the three frames concatenated the way script/giremi.py:79-88 concatenates them.
    ``threads`` > 1 with ``reopen`` (a picklable callable returning fresh ``(sam, genome)`` objects) runs the host-side
    extraction in a process pool; ``engine`` may then be a callable: it is called once the workers are forked (a HIP
    context does not survive fork()).  With ``concat`` on one rank the run is a PIPELINE (round 4): the chunks of
    footprints come back from the pool in order, and while the workers extract the later ones the parent runs each
    chunk's own batch on the GPU (``stream_site_base`` = the sites before it: pair for pair the permutation draws of the one
    batch holding every footprint), builds its rows of the two site tables and hands its part of the removed-site table to
    ``removed_sink`` (a callable taking a DataFrame; the returned df_removed is then empty); ``pairs_sink`` (a callable taking a
    chunk's pair rows as a DataFrame, in order, and its string columns as dictionary codes — regions_pair_mi_table: codes_out);
    ``pool``: a multiprocessing pool of forked workers to use instead of making one (the caller forked it while still small) lets the caller write the pair table while the run goes on — the returned
    df_pairs is still the whole table."""
    import time
    t0 = time.perf_counter()
    footprints = list(footprints)
    staged = None
    prepacked = None                        # the footprints' blocks already packed by the extraction workers
    multi_rank = group is not None and group.get_world_size() > 1
    if threads and threads > 1 and len(footprints) > 1 and reopen is not None:
        # host-side site extraction is per footprint and shares nothing: the reference maps it over a process pool
        # (script/giremi.py:367-380, -t); so does this, with the MI step kept OUT of the workers — they return site
        # dictionaries and their footprints' packed blocks.  Workers reopen the files (`reopen()` -> (sam, genome)); the
        # parent creates its HIP context only after the workers exist (fork).
        import multiprocessing as mp
        chunk = max(1, -(-len(footprints) // (4 * threads)))
        lean = bool(concat and not multi_rank)                  # (the pipelined run below)
        jobs = [(reopen, footprints[k:k + chunk], filter_kwargs, bool(concat), bool(concat), lean)
                for k in range(0, len(footprints), chunk)]
        import contextlib
        # `pool`: the caller's pool of forked workers (made before it loaded its inputs: lgmi.cli) — used, not closed
        with (contextlib.nullcontext(pool) if pool is not None else mp.get_context('fork').Pool(threads)) as pool:
            if concat and not multi_rank:
                parts = _ordered_parts(pool, _extract_chunk, jobs)    # in job order, as they finish
                if callable(engine) and not hasattr(engine, 'run'):
                    engine = engine()
                t_gpu = t_tab = 0.0
                site_base, site_rows, pair_frames, removed_frames = 0, [], [], []
                chrom_code, reason_code = {}, {}
                nan = float('nan')
                trace_jobs = [] if os.environ.get('LGMI_TRACE_JOBS') else None
                for part, b, extras in parts:
                    t1 = time.perf_counter()
                    if trace_jobs is not None:
                        trace_jobs.append(extras.get('job', (0, 0.0, 0.0)) + (time.time(),))
                    kw = {'site_base': site_base} if site_base else {}
                    codes = {} if pairs_sink is not None else None
                    df_c, means_c = regions_pair_mi_table([(sites, chrom) for chrom, sites, _gone in part], min_common_reads,
                                                          n_shuffles=n_shuffles, seed=seed, engine=engine, batch=b, codes_out=codes, **kw)
                    site_base += len(b.site_pos) if b is not None else 0
                    pair_frames.append(df_c)
                    if pairs_sink is not None and len(df_c):
                        pairs_sink(df_c, codes)
                    t2 = time.perf_counter()
                    # the worker made the footprints' rows of the mismatch table; their last column is the GPU's per-site mean
                    for rows, mean_mi in zip(extras['site_rows'], means_c):
                        for row in rows:
                            row[12] = mean_mi[row[2]].get(row[3], nan)
                        site_rows.extend(rows)
                    if removed_sink is not None and hasattr(removed_sink, 'write_arrays'):
                        removed_sink.write_arrays(extras['removed'])          # (formatted natively: no frame is made)
                    else:
                        gone_frame = _removed_frame_from_arrays(extras['removed'], chrom_code if removed_sink is None else {},
                                                                reason_code if removed_sink is None else {})
                        if removed_sink is not None:
                            removed_sink(gone_frame)
                        else:
                            removed_frames.append(gone_frame)
                    t_gpu += t2 - t1
                    t_tab += time.perf_counter() - t2
                if removed_sink is not None:
                    df_removed = _removed_frame([], {}, {})
                elif len(removed_frames) == 1:
                    df_removed = removed_frames[0]
                else:                                                  # (categories grew from chunk to chunk: the last frame's hold them all)
                    from pandas.api.types import union_categoricals
                    df_removed = pd.DataFrame({c: (union_categoricals([f[c] for f in removed_frames]) if c != 'pos'
                                                   else np.concatenate([f[c].to_numpy() for f in removed_frames]))
                                               for c in ('chromosome', 'strand', 'pos', 'removed')}) if removed_frames \
                        else _removed_frame([], {}, {})
                if trace_jobs:
                    import sys
                    base = min(j[1] for j in trace_jobs)
                    for k, (pid, a, b_, got) in enumerate(trace_jobs):
                        print('[lgmi jobs] %3d pid %d ran %.3f - %.3f s, received %.3f s' % (k, pid, a - base, b_ - base, got - base), file=sys.stderr)
                if timing is not None:
                    timing['extract_s'] = time.perf_counter() - t0          # the pipeline's wall time: extraction with the
                    timing['pack_gpu_table_s'] = t_gpu                       # parent's GPU and table work (listed beside it)
                    timing['site_tables_s'] = t_tab                          # running underneath
                    timing['pipelined'] = True
                full = [f for f in pair_frames if len(f)] or pair_frames[:1]       # (an empty frame has no dtypes to give)
                df_pairs = pd.concat(full, axis=0, ignore_index=True) if len(full) > 1 else full[0]
                return pd.DataFrame.from_records(site_rows, columns=_SITE_COLS), df_pairs, df_removed
            parts = pool.map(_extract_chunk, jobs)
        if concat:
            from .pack import concat_batches
            staged = [x for part, _b, _extras in parts for x in part]
            batches = [b for _part, b, _extras in parts if b is not None]
            prepacked = concat_batches(batches) if batches else None
        else:
            staged = [x for part in parts for x in part]
    if staged is None:
        staged = _extract_chunk((None, footprints, filter_kwargs, bool(concat)), sam, genome)
    if callable(engine) and not hasattr(engine, 'run'):
        engine = engine()
    if timing is not None:
        timing['extract_s'] = time.perf_counter() - t0
        t0 = time.perf_counter()
    if concat:
        # a whole run: the pair table column by column from the result arrays, the two site tables from one row list each
        # (a DataFrame per footprint and a concat of thousands of them was most of the host time on 2,000 footprints)
        if multi_rank:
            # several ranks, each with its own contiguous run of footprints: the pair rows are gathered over RCCL onto
            # rank 0 (df_pairs is None elsewhere); the two site tables stay per rank (the caller concatenates them)
            df_pairs, means = regions_pair_mi_table_dist([(sites, chrom) for chrom, sites, _gone in staged], group, engine,
                                                         min_common_reads, n_shuffles=n_shuffles, seed=seed, batch=prepacked)
        else:
            df_pairs, means = regions_pair_mi_table([(sites, chrom) for chrom, sites, _gone in staged], min_common_reads,
                                                    n_shuffles=n_shuffles, seed=seed, engine=engine, batch=prepacked)
        if timing is not None:
            timing['pack_gpu_table_s'] = time.perf_counter() - t0
            t0 = time.perf_counter()
        site_rows = []
        for (chrom, sites, _gone), mean_mi in zip(staged, means):
            site_rows.extend(_site_rows(chrom, sites, mean_mi))
        df_removed = _removed_frame(staged, {}, {})
        if removed_sink is not None:
            removed_sink(df_removed)
            df_removed = _removed_frame([], {}, {})
        out = (pd.DataFrame.from_records(site_rows, columns=_SITE_COLS), df_pairs, df_removed)
        if timing is not None:
            timing['site_tables_s'] = time.perf_counter() - t0
        return out
    blocks = regions_pair_mi([(sites, chrom) for chrom, sites, _gone in staged], min_common_reads,
                             n_shuffles=n_shuffles, seed=seed, engine=engine)
    return [_frames(chrom, sites, gone, *blk) for (chrom, sites, gone), blk in zip(staged, blocks)]


_SITE_COLS = ['type', 'chromosome', 'strand', 'pos', 'ref', 'change_type', 'ratio', 'allelic_ratio_diff',
              'depth', 'A:C:T:G', 'up_seq', 'down_seq', 'mean_mi']


def _site_rows(chromosome, sites, mean_mi):
    """rows of the mismatch table (mismatch.py:419-495) of one footprint"""
    rows = []
    for strand in '+-':
        means = mean_mi[strand]

        def alt_ratio(site):
            depth = site['depth']
            total = sum(depth[nt] for nt in depth)
            alts = sorted(([nt, depth[nt]] for nt in depth if nt != site['ref']), key=lambda t: t[1], reverse=True)
            return alts[0][0], alts[0][1] / total, total

        hets = [alt_ratio(s)[1] for s in sites[strand].values() if s['type'] == 'het_snp']
        regional = sum(hets) / len(hets) if hets else 0.5           # :424-443
        for pos, site in sites[strand].items():
            alt, ratio, total = alt_ratio(site)
            depth = dict(site['depth'])
            ref = site['ref']
            if strand == '+':
                change, up, down = '%s>%s' % (ref, alt), site['up'], site['down']
            else:
                change = '%s>%s' % (_COMPLEMENT[ref], _COMPLEMENT[alt])
                up, down = _COMPLEMENT[site['down']], _COMPLEMENT[site['up']]
            if 'N' in change:
                continue
            acgt = '%s:%s:%s:%s' % tuple(depth.get(nt, 0) for nt in 'ACTG')     # A:C:T:G, defaultdict zeros
            rows.append([site['type'], chromosome, strand, pos, ref, change, ratio, ratio - regional, total,
                         acgt, up, down, means.get(pos, np.nan)])
    return rows


def _frames(chromosome, sites, gone, records, mean_mi, pvals):
    """the three DataFrames of mismatch.py:406-509 from the filtered sites and the MI block's output"""
    pair_cols = ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
    df_pairs = pd.DataFrame.from_records(records, columns=pair_cols)
    if pvals is not None:
        df_pairs['p_perm'] = pvals

    df_sites = pd.DataFrame.from_records(_site_rows(chromosome, sites, mean_mi), columns=_SITE_COLS)
    df_removed = pd.DataFrame.from_records(
        [[chromosome, s, pos, gone[s][pos]['removed']] for s in '+-' for pos in gone[s]],
        columns=['chromosome', 'strand', 'pos', 'removed'])
    return df_sites, df_pairs, df_removed
