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
                          p(self.site_plane_off, _lib.u64p), p(self.planes, _lib.u64p))

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
                   np.zeros(ns, bool))


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
                       np.concatenate([p.bad_sites if p.bad_sites is not None else np.zeros(p.n_sites, bool) for p in parts]))


def _site_classes(site: dict, read_index: Dict[str, int]):
    """(read indices, classes) of one site; new read names get the next index."""
    read_allele: Dict[str, str] = {}
    for allele, names in site['nt'].items():
        for name in names:
            read_allele[name] = allele                      # last allele wins (:15-16)
    ranked = sorted(site['depth'].items(), key=lambda kv: kv[1], reverse=True)   # stable (:27-28)
    bad = len(ranked) < 2                                   # the reference raises IndexError (:30,:32)
    major = ranked[0][0] if ranked else None
    minor = ranked[1][0] if len(ranked) > 1 else None
    idx = np.empty(len(read_allele), np.int64)
    cls = np.empty(len(read_allele), np.uint8)
    for k, (name, allele) in enumerate(read_allele.items()):
        r = read_index.get(name)
        if r is None:
            r = len(read_index)
            read_index[name] = r
        idx[k] = r
        cls[k] = 2 if allele == major else (1 if allele == minor else 0)
    return idx, cls, bad


def pack_blocks(blocks: Sequence[dict]) -> PackedBatch:
    """one block per ``mismatches[strand]`` dict (positions -> site dict)."""
    bsb = [0]
    n_reads, pos, typ, names, woff, nwords, poff, chunks, bad = [], [], [], [], [], [], [], [], []
    total = 0
    one = np.uint64(1)
    for mismatches in blocks:
        read_index: Dict[str, int] = {}
        for p in sorted(mismatches.keys()):
            site = mismatches[p]
            idx, cls, is_bad = _site_classes(site, read_index)
            if idx.size:
                w0, w1 = int(idx.min()) >> 6, (int(idx.max()) >> 6) + 1
            else:
                w0, w1 = 0, 0
            nw = w1 - w0
            lo = np.zeros(nw, np.uint64)
            hi = np.zeros(nw, np.uint64)
            if idx.size:
                word = (idx >> 6) - w0
                bit = np.left_shift(one, (idx & 63).astype(np.uint64))
                np.bitwise_or.at(lo, word[cls != 2], bit[cls != 2])   # class 1 and class 0 set lo
                np.bitwise_or.at(hi, word[cls != 1], bit[cls != 1])   # class 2 and class 0 set hi
            pos.append(int(p))
            t = site['type']
            names.append(t)
            typ.append(TYPE_CODE.get(t, _lib.TYPE_MISMATCH))
            woff.append(w0)
            nwords.append(nw)
            poff.append(total)
            chunks.append(lo)
            chunks.append(hi)
            total += 2 * nw
            bad.append(is_bad)
        bsb.append(len(pos))
        n_reads.append(len(read_index))
    planes = np.concatenate(chunks) if chunks else np.zeros(0, np.uint64)
    return PackedBatch(np.asarray(bsb, np.uint64), np.asarray(n_reads, np.uint32), np.asarray(pos, np.int64),
                       np.asarray(typ, np.uint8), np.asarray(woff, np.uint32), np.asarray(nwords, np.uint32),
                       np.asarray(poff, np.uint64), np.ascontiguousarray(planes, np.uint64), names,
                       np.asarray(bad, bool))
