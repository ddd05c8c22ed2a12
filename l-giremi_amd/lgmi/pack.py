"""Host-side packer: the reference's ``mismatches[strand]`` dicts -> packed blocks
(include/lgmi.h ``lgmi_batch``).

Reference semantics reproduced at pack time (src/giremi/mutual_information.py):
  :10-11  sites sorted by position
  :15-16  a read listed under several alleles keeps the LAST allele in ``nt`` order
  :25-40  alleles ranked by the site-wide ``depth`` dict (descending, stable);
          rank 0 -> class 2, rank 1 -> class 1, all others -> class 0
Because the class of a read at a site does not depend on the partner site, it is
assigned once here and the kernels are label-agnostic.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List, Sequence

import numpy as np

from . import _lib

TYPE_CODE = {'mismatch': _lib.TYPE_MISMATCH, 'snp': _lib.TYPE_SNP, 'het_snp': _lib.TYPE_HET_SNP}


@dataclass
class PackedBatch:
    """numpy arrays behind an ``lgmi_batch``; ``sites[k]`` lists (pos, type string) of block k."""
    block_site_begin: np.ndarray
    block_n_reads: np.ndarray
    site_pos: np.ndarray
    site_type: np.ndarray
    site_word_off: np.ndarray
    site_n_words: np.ndarray
    site_plane_off: np.ndarray
    planes: np.ndarray
    type_names: List[str] = field(default_factory=list)   # original type string of every site
    bad_sites: np.ndarray = None                          # sites whose depth dict has < 2 alleles
    site_tri: np.ndarray = None                           # uint8 [n_sites]: the site has class-0 reads (lgmi_batch.site_tri), or None

    @property
    def n_blocks(self):
        return len(self.block_n_reads)

    @property
    def n_sites(self):
        return len(self.site_pos)

    def as_struct(self) -> _lib.Batch:
        def p(a, t):
            return a.ctypes.data_as(t) if a.size else C.cast(None, t)
        return _lib.Batch(self.n_blocks, self.n_sites, self.planes.size,
                          p(self.block_site_begin, _lib.u64p), p(self.block_n_reads, _lib.u32p),
                          p(self.site_pos, _lib.i64p), p(self.site_type, _lib.u8p),
                          p(self.site_word_off, _lib.u32p), p(self.site_n_words, _lib.u32p),
                          p(self.site_plane_off, _lib.u64p), p(self.planes, _lib.u64p),
                          p(self.site_tri, _lib.u8p) if self.site_tri is not None and len(self.site_tri) == self.n_sites
                          else C.cast(None, _lib.u8p))

    @classmethod
    def from_struct(cls, b: _lib.Batch) -> 'PackedBatch':
        """deep copy of a library-owned batch (lgmi_dbatch_download)"""
        def a(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        ns, nb = int(b.n_sites), int(b.n_blocks)
        types = a(b.site_type, ns, np.uint8)
        names = ['mismatch', 'snp', 'het_snp']
        return cls(a(b.block_site_begin, nb + 1, np.uint64), a(b.block_n_reads, nb, np.uint32),
                   a(b.site_pos, ns, np.int64), types, a(b.site_word_off, ns, np.uint32),
                   a(b.site_n_words, ns, np.uint32), a(b.site_plane_off, ns, np.uint64),
                   a(b.planes, int(b.n_plane_words), np.uint64), [names[t] for t in types],
                   np.zeros(ns, bool), a(b.site_tri, ns, np.uint8) if b.site_tri else None)


def concat_batches(parts: Sequence['PackedBatch']) -> 'PackedBatch':
    """several packed batches as one (blocks keep their order; site and plane offsets are shifted)"""
    bsb, poff, site0, plane0 = [np.zeros(1, np.uint64)], [], 0, 0
    for p in parts:
        bsb.append(p.block_site_begin[1:].astype(np.uint64) + np.uint64(site0))
        poff.append(p.site_plane_off.astype(np.uint64) + np.uint64(plane0))
        site0 += p.n_sites
        plane0 += p.planes.size
    cat = lambda f, dt: np.concatenate([getattr(p, f) for p in parts]).astype(dt)
    return PackedBatch(np.concatenate(bsb), cat('block_n_reads', np.uint32), cat('site_pos', np.int64),
                       cat('site_type', np.uint8), cat('site_word_off', np.uint32), cat('site_n_words', np.uint32),
                       np.concatenate(poff), cat('planes', np.uint64), sum((list(p.type_names) for p in parts), []),
                       np.concatenate([p.bad_sites if p.bad_sites is not None else np.zeros(p.n_sites, bool) for p in parts]),
                       np.concatenate([p.site_tri for p in parts]).astype(np.uint8) if all(p.site_tri is not None for p in parts) else None)


def _or_scatter(planes: np.ndarray, target: np.ndarray, bits: np.ndarray):
    """planes[target] |= bits for a SORTED target array: equal targets form runs, one bitwise_or.reduceat folds them"""
    if target.size == 0:
        return
    starts = np.flatnonzero(np.concatenate(([True], target[1:] != target[:-1])))
    planes[target[starts]] = np.bitwise_or.reduceat(bits, starts)


def _pack_one(mismatches: dict):
    """one block, vectorised: the only per-read Python work is list.extend of the name lists; read numbering,
    duplicate handling, band limits and the two bit planes are array operations.
    -> (pos, type names, bad flags, n_reads, word_off, n_words, planes of the block, plane offsets inside it, tri flags)"""
    positions = sorted(mismatches.keys())
    P = len(positions)
    names_flat: List[str] = []
    run_len, run_site, run_cls, type_names = [], [], [], []
    bad = np.zeros(P, bool)
    for si, p in enumerate(positions):
        site = mismatches[p]
        type_names.append(site['type'])
        ranked = sorted(site['depth'].items(), key=lambda kv: kv[1], reverse=True)   # stable (:27-28)
        bad[si] = len(ranked) < 2                               # the reference raises IndexError (:30,:32)
        major = ranked[0][0] if ranked else None
        minor = ranked[1][0] if len(ranked) > 1 else None
        for allele, names in site['nt'].items():
            names_flat.extend(names)
            run_len.append(len(names))
            run_site.append(si)
            run_cls.append(2 if allele == major else (1 if allele == minor else 0))
    pos = np.asarray(positions, np.int64)
    n = len(names_flat)
    if n == 0:
        z = np.zeros(P, np.int64)
        return pos, type_names, bad, 0, z, z, np.zeros(0, np.uint64), z, np.zeros(P, np.uint8)
    # reads are numbered in order of first appearance (sites by position, alleles in `nt` order): what factorize does
    import pandas as pd
    # (read "names" may be integer read ids — lgmi.region's native extraction with read_ids: same identity, no strings)
    code, uniques = pd.factorize(np.asarray(names_flat, np.int64) if isinstance(names_flat[0], int) else np.asarray(names_flat, dtype=object))
    R = len(uniques)
    site_of = np.repeat(np.asarray(run_site, np.int64), run_len)
    cls_of = np.repeat(np.asarray(run_cls, np.uint8), run_len)
    # a read listed under several alleles of a site keeps the LAST one (:15-16): first occurrence in reversed order
    key = site_of * R + code.astype(np.int64)
    ukey, ridx = np.unique(key[::-1], return_index=True)        # sorted by (site, read)
    cls_u = cls_of[n - 1 - ridx]
    site_u, read_u = ukey // R, ukey % R
    word = read_u >> 6
    first = np.searchsorted(site_u, np.arange(P), side='left')
    last = np.searchsorted(site_u, np.arange(P), side='right')
    has = last > first
    # reads of a site are NOT sorted by index inside ukey?  they are: ukey is sorted and site-major, so read_u increases
    w0 = np.where(has, word[np.minimum(first, len(word) - 1)], 0)
    w1 = np.where(has, word[np.maximum(last - 1, 0)] + 1, 0)
    nw = w1 - w0
    poff = 2 * (np.cumsum(nw) - nw)
    planes = np.zeros(int(2 * nw.sum()), np.uint64)
    bit = np.left_shift(np.uint64(1), (read_u & 63).astype(np.uint64))
    lo_at = poff[site_u] + (word - w0[site_u])
    m_lo, m_hi = cls_u != 2, cls_u != 1                          # class 1 and class 0 set lo; class 2 and class 0 set hi
    _or_scatter(planes, lo_at[m_lo], bit[m_lo])
    _or_scatter(planes, (lo_at + nw[site_u])[m_hi], bit[m_hi])
    tri = (np.bincount(site_u[cls_u == 0], minlength=P) > 0).astype(np.uint8)     # lgmi_batch.site_tri: the site has class-0 reads
    return pos, type_names, bad, R, w0, nw, planes, poff, tri


def pack_blocks(blocks: Sequence[dict]) -> PackedBatch:
    """one block per ``mismatches[strand]`` dict (positions -> site dict)."""
    bsb, n_reads, pos, names, woff, nwords, poff, chunks, bad, tri = [0], [], [], [], [], [], [], [], [], []
    total = 0
    for mismatches in blocks:
        p, tn, b, R, w0, nw, planes, po, tr = _pack_one(mismatches)
        pos.append(p); names.extend(tn); bad.append(b); woff.append(w0); nwords.append(nw); tri.append(tr)
        poff.append(po + total)
        chunks.append(planes)
        total += planes.size
        bsb.append(bsb[-1] + len(p))
        n_reads.append(R)

    def cat(parts, dt):
        return np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    typ = np.asarray([TYPE_CODE.get(t, _lib.TYPE_MISMATCH) for t in names], np.uint8)
    return PackedBatch(np.asarray(bsb, np.uint64), np.asarray(n_reads, np.uint32), cat(pos, np.int64), typ,
                       cat(woff, np.uint32), cat(nwords, np.uint32), cat(poff, np.uint64),
                       np.ascontiguousarray(cat(chunks, np.uint64)), names, cat(bad, bool), cat(tri, np.uint8))
