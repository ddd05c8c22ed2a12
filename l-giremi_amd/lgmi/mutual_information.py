"""Drop-ins for the reference's site-pair MI step, computed on the MI355X.

Same names, arguments, return shapes and error behaviour as
``src/giremi/mutual_information.py`` of gxiaolab/L-GIREMI (v0.2.4):

    mismatch_pair_mutual_info(mismatches, min_common_reads=5)   (:6-45)
    mean_mismatch_pair_mutual_info(mismatch_pair_mi)            (:48-60)

plus ``region_pair_mi`` — the batched form of the caller's MI block
(``src/giremi/mismatch.py:384-418``) that does both strands in one launch
sequence and is what a GPU-aware host should call.

Everything numeric happens in liblgmi.so (HIP, gfx950) through ctypes; there is
no CPU fallback — without the library or a GPU these functions raise.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from .engine import Engine, default_engine
from .pack import pack_blocks


def _min_common(min_common_reads) -> int:
    # the reference tests len(common) < min_common_reads (:19); any real threshold works
    return max(0, int(math.ceil(min_common_reads)))


def mismatch_pair_mutual_info(mismatches: dict, min_common_reads=5, engine: Optional[Engine] = None) -> List[list]:
    """All pairs of positions (sorted, ``itertools.combinations`` order) with at least
    ``min_common_reads`` common reads -> ``[p1, type1, p2, type2, mi]`` rows."""
    if len(mismatches) < 2:
        return []
    eng = engine or default_engine()
    batch = pack_blocks([mismatches])
    res = eng.run(batch, min_common=_min_common(min_common_reads), het_only=False)
    if batch.bad_sites.any():
        # the reference ranks the alleles of BOTH sites of every qualifying pair — het-involved or not — before the
        # het filter (mutual_information.py:25-32, mismatch.py:392-396): a site whose depth dict has < 2 alleles raises
        # IndexError as soon as it is in any pair with enough common reads.  `res` already is the all-pairs result
        # (het_only=False), so it answers that directly.
        if res.n_rows and (batch.bad_sites[res.row_i].any() or batch.bad_sites[res.row_j].any()):
            raise IndexError('list index out of range')   # fewer than two alleles in depth (:30,:32)
    pos, names = batch.site_pos, batch.type_names
    keys = sorted(mismatches.keys())                       # hand back the caller's own key objects
    return [[keys[i], names[i], keys[j], names[j], float(mi)]
            for i, j, mi in zip(res.row_i.tolist(), res.row_j.tolist(), res.row_mi.tolist())]


def mean_mismatch_pair_mutual_info(mismatch_pair_mi, engine: Optional[Engine] = None) -> List[list]:
    """``[pos, mean mi]`` per site, in order of first appearance; every row counts
    for both of its sites."""
    index: Dict[object, int] = {}
    ri, rj, rm = [], [], []
    for p1, _t1, p2, _t2, mi in mismatch_pair_mi:
        ri.append(index.setdefault(p1, len(index)))
        rj.append(index.setdefault(p2, len(index)))
        rm.append(mi)
    if not index:
        return []
    eng = engine or default_engine()
    mean, _cnt = eng.site_mean(ri, rj, rm, len(index))
    return [[pos, float(mean[k])] for pos, k in index.items()]


def _run_regions(regions, min_common_reads, n_shuffles, seed, engine, batch=None, site_base=0):
    """pack every (footprint, strand) of `regions` as one batch and run it -> (batch, result) or (None, None);
    `batch`: the same blocks already packed (the extraction workers of a whole run pack their own footprints)"""
    blocks = []
    for mm, _chrom in regions:
        blocks.extend(mm.get(s, {}) for s in ('+', '-'))
    if not any(len(b) > 1 for b in blocks):
        return None, None
    eng = engine or default_engine()
    if batch is None:
        batch = pack_blocks(blocks)
    elif batch.n_blocks != len(blocks):
        raise ValueError('pre-packed batch has %d blocks for %d (footprint, strand) pairs' % (batch.n_blocks, len(blocks)))
    res = eng.run(batch, min_common=_min_common(min_common_reads), n_shuffles=n_shuffles, seed=seed, het_only=True,
                  stream_site_base=site_base)
    if batch.bad_sites.any():
        # the reference ranks the alleles of BOTH sites of every qualifying pair — het-involved or not — before the
        # het filter (mutual_information.py:25-32, mismatch.py:392-396): a site whose depth dict has < 2 alleles raises
        # IndexError as soon as it is in any pair with enough common reads.  Such sites never reach this point in the
        # pipeline (mismatch.py:275-282), so the extra all-pairs pass only runs for hand-made inputs.
        chk = eng.run(batch, min_common=_min_common(min_common_reads), het_only=False)
        if chk.n_rows and (batch.bad_sites[chk.row_i].any() or batch.bad_sites[chk.row_j].any()):
            raise IndexError('list index out of range')
    return batch, res


def _site_means(batch, res, n_regions):
    """per region {'+': {pos: mean}, '-': {pos: mean}} from the batch's per-site figures"""
    strands = ('+', '-')
    out = [{'+': {}, '-': {}} for _ in range(n_regions)]
    bsb = batch.block_site_begin.astype(np.int64)
    pos = batch.site_pos.tolist()
    npairs, mean = res.site_n_pairs, res.site_mean_mi
    hit = np.nonzero(npairs)[0]
    for s, b in zip(hit.tolist(), (np.searchsorted(bsb, hit, side='right') - 1).tolist()):
        out[b >> 1][strands[b & 1]][pos[s]] = float(mean[s])
    return out


def regions_pair_mi(regions, min_common_reads=5, n_shuffles=0, seed=0, engine: Optional[Engine] = None):
    """The MI block (mismatch.py:384-418) of MANY footprints in ONE launch sequence.

    ``regions`` is a sequence of ``(mismatches_by_strand, chromosome)``; every (footprint, strand) becomes one
    block of a single batch, so thousands of small footprints cost one upload, one count launch and one
    fetch instead of thousands.  Returns one ``(records, mean_mi, p_values)`` triple per region, each exactly
    what ``region_pair_mi`` returns for that region alone."""
    strands = ('+', '-')
    regions = list(regions)
    out = [([], {'+': {}, '-': {}}, ([] if n_shuffles else None)) for _ in regions]
    batch, res = _run_regions(regions, min_common_reads, n_shuffles, seed, engine)
    if batch is None:
        return out
    bsb = batch.block_site_begin.astype(np.int64)
    pos, names = batch.site_pos.tolist(), batch.type_names
    # rows come sorted by block: one slice per block instead of a Python step per row
    cut = np.searchsorted(res.row_i.astype(np.int64), bsb, side='left')
    ri, rj, rmi = res.row_i.tolist(), res.row_j.tolist(), res.row_mi.tolist()
    rp = res.row_p.tolist() if n_shuffles else None
    for b in range(len(bsb) - 1):
        r0, r1 = int(cut[b]), int(cut[b + 1])
        if r0 == r1:
            continue
        records, _means, pvals = out[b >> 1]
        chrom, strand = regions[b >> 1][1], strands[b & 1]
        records.extend([chrom, strand, pos[i], names[i], pos[j], names[j], m]
                       for i, j, m in zip(ri[r0:r1], rj[r0:r1], rmi[r0:r1]))
        if pvals is not None:
            pvals.extend(rp[r0:r1])
    for k, m in enumerate(_site_means(batch, res, len(regions))):
        out[k][1].update(m)
    return out


def regions_pair_mi_table(regions, min_common_reads=5, n_shuffles=0, seed=0, engine: Optional[Engine] = None, batch=None,
                          site_base=0, codes_out=None):
    """The same launch, returned the way a whole run consumes it: ONE table of all regions' pair rows in the
    reference's order (script/giremi.py:381-394 concatenates the per-footprint frames of mismatch.py:407-418) built
    column by column from the result arrays — no Python object per row, which is what 10^7 rows of tens of
    thousands of footprints need — plus the per-region ``mean_mi`` dictionaries.
    ``site_base``: the number of sites of the run that precede this batch (a run fed to the GPU chunk by chunk): added to
    the site indices in the permutation draws' counters, so that the chunks draw what the one batch of all of them draws.
    ``codes_out``: a dict that receives the table's string columns once more as dictionary codes, column -> (int32 codes,
    names) — what a native table writer takes (lgmi.io.write_table; the CLI's PREFIX.mi.txt: formatting 200,000 rows through
    to_csv was 0.7 s of the pipeline's parent).
    -> (DataFrame[chromosome, strand, site1_pos, site1_type, site2_pos, site2_type, mi (, p_perm)], [mean_mi])"""
    import pandas as pd
    regions = list(regions)
    cols = ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
    batch, res = _run_regions(regions, min_common_reads, n_shuffles, seed, engine, batch, site_base)
    if batch is None or res.n_rows == 0:
        df = pd.DataFrame({c: [] for c in cols})
        if n_shuffles:
            df['p_perm'] = []
        return df, ([{'+': {}, '-': {}} for _ in regions] if batch is None else _site_means(batch, res, len(regions)))
    bsb = batch.block_site_begin.astype(np.int64)
    block = np.searchsorted(bsb, res.row_i.astype(np.int64), side='right') - 1          # the block of every row
    chrom_of_region = np.array([c for _mm, c in regions], dtype=object)
    names = np.array(batch.type_names, dtype=object)
    pos = batch.site_pos
    row_i, row_j = res.row_i, res.row_j
    df = pd.DataFrame({'chromosome': chrom_of_region[block >> 1],
                       'strand': np.array(['+', '-'], dtype=object)[block & 1],
                       'site1_pos': pos[row_i], 'site1_type': names[row_i],
                       'site2_pos': pos[row_j], 'site2_type': names[row_j],
                       'mi': res.row_mi}, columns=cols)
    if n_shuffles:
        df['p_perm'] = res.row_p
    if codes_out is not None:
        type_names, type_code = np.unique(np.array(batch.type_names, dtype=str), return_inverse=True)
        chrom_names, chrom_code = np.unique(np.array([str(c) for _mm, c in regions], dtype=str), return_inverse=True)
        codes_out.update({'chromosome': (chrom_code[block >> 1].astype(np.int32), list(chrom_names)),
                          'strand': ((block & 1).astype(np.int32), ['+', '-']),
                          'site1_type': (type_code[row_i].astype(np.int32), list(type_names)),
                          'site2_type': (type_code[row_j].astype(np.int32), list(type_names))})
    return df, _site_means(batch, res, len(regions))


def regions_pair_mi_table_dist(regions, group, engine: Engine, min_common_reads=5, n_shuffles=0, seed=0, root=0, batch=None):
    """regions_pair_mi_table when the footprints of a run were dealt to several ranks (one process per GPU; the
    reference's analogue is the chunked Pool.map over footprints, src/giremi/script/giremi.py:367-394): every rank
    packs and runs ITS regions, the pair rows of all ranks travel HBM-to-HBM onto `root` in rank order
    (lgmi_comm_gather: RCCL, site indices shifted by each rank's site base), the site metadata the table needs
    (positions, types, block boundaries, chromosome of every region) over the host-side group.  Ranks hold contiguous
    runs of footprints, so rank order is footprint order.  -> (table on root / None elsewhere, this rank's [mean_mi])"""
    import pandas as pd
    from .pack import PackedBatch
    regions = list(regions)
    rank, world = group.get_rank(), group.get_world_size()
    blocks = []
    for mm, _chrom in regions:
        blocks.extend(mm.get(s, {}) for s in ('+', '-'))
    if batch is not None and blocks and batch.n_blocks != len(blocks):
        # (as _run_regions: the workers that packed the batch have replaced the read lists by counts — repacking from `regions`
        #  here would silently pack empty sites)
        raise ValueError('pre-packed batch has %d blocks for %d (footprint, strand) pairs' % (batch.n_blocks, len(blocks)))
    if batch is None or not blocks:
        batch = pack_blocks(blocks) if blocks else pack_blocks([{}])
    n_sites = len(batch.site_pos)
    counts = group.allgather(int(n_sites))
    base = int(sum(counts[:rank]))
    db = engine.upload(batch)
    dr = engine.run_device(db, min_common=_min_common(min_common_reads), n_shuffles=n_shuffles, seed=seed, het_only=True,
                           stream_site_base=base)       # pair for pair the permutation draws of the one batch holding every footprint
    local = dr.fetch()                                   # per-site means of this rank's own footprints (small)
    if blocks and batch.bad_sites.any():
        # the single-rank path's parity check (_run_regions): a site with < 2 alleles in a qualifying pair is the reference's
        # IndexError (mutual_information.py:25-32); every rank checks its own footprints
        chk = engine.run(batch, min_common=_min_common(min_common_reads), het_only=False)
        if chk.n_rows and (batch.bad_sites[chk.row_i].any() or batch.bad_sites[chk.row_j].any()):
            raise IndexError('list index out of range')
    gathered, _begins = engine.comm_gather(dr, root=root, site_base=base, same_batch=False)
    meta = group.gather({'pos': batch.site_pos, 'types': list(batch.type_names),
                         'bsb': batch.block_site_begin.astype(np.int64), 'chroms': [c for _mm, c in regions]}, root)
    means = _site_means(batch, local, len(regions)) if blocks else []
    table = None
    if rank == root:
        g = gathered.fetch()
        gathered.free()
        pos = np.concatenate([np.asarray(m['pos'], np.int64) for m in meta]) if meta else np.zeros(0, np.int64)
        names = np.array([t for m in meta for t in m['types']], dtype=object)
        # global block boundaries and the chromosome of every region, in rank order
        bsb, chroms, off = [np.zeros(1, np.int64)], [], 0
        for m in meta:
            b = np.asarray(m['bsb'], np.int64)
            if len(m['chroms']):
                bsb.append(b[1:] + off)
                chroms.extend(m['chroms'])
            off += len(m['pos'])
        bsb = np.concatenate(bsb)
        cols = ['chromosome', 'strand', 'site1_pos', 'site1_type', 'site2_pos', 'site2_type', 'mi']
        if g.n_rows:
            block = np.searchsorted(bsb, g.row_i.astype(np.int64), side='right') - 1
            table = pd.DataFrame({'chromosome': np.array(chroms, dtype=object)[block >> 1],
                                  'strand': np.array(['+', '-'], dtype=object)[block & 1],
                                  'site1_pos': pos[g.row_i], 'site1_type': names[g.row_i],
                                  'site2_pos': pos[g.row_j], 'site2_type': names[g.row_j], 'mi': g.row_mi}, columns=cols)
            if n_shuffles:
                table['p_perm'] = g.row_p
        else:
            table = pd.DataFrame({c: [] for c in cols})
            if n_shuffles:
                table['p_perm'] = []
    dr.free()
    db.free()
    return table, means


def region_pair_mi(mismatches_by_strand: dict, chromosome: str, min_common_reads=5, n_shuffles=0, seed=0,
                   engine: Optional[Engine] = None):
    """Batched MI block of ``region_mismatch_analysis`` (mismatch.py:384-418) for one footprint.

    Returns ``(records, mean_mi, p_values)``: ``records`` are the rows of
    ``df_mismatch_pair_mi`` — ``[chromosome, strand, p1, type1, p2, type2, mi]``, '+'
    rows then '-' rows, het_snp-involved pairs only; ``mean_mi[strand]`` maps position
    -> mean MI (what ``mean_mismatch_pair_mutual_info`` feeds the mismatch table);
    ``p_values`` is None unless ``n_shuffles`` > 0 (then one permutation p per record).
    """
    return regions_pair_mi([(mismatches_by_strand, chromosome)], min_common_reads, n_shuffles, seed, engine)[0]
