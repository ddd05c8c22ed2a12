"""Sequence-file access for the l-giremi-compatible CLI.

The reference opens its inputs with pysam (src/giremi/script/giremi.py:21-24); pysam/htslib cannot be installed
here, so this module provides the pysam attributes that the path actually touches (src/giremi/mismatch.py:69-190,
src/giremi/footprint.py:6-28, src/giremi/fileio.py:24-30):

    BamReader   ~ pysam.AlignmentFile   fetch(), pileup(); reads expose query_name, reference_start,
                                        reference_end, is_reverse, get_tag() ...  A thin ctypes wrapper over
                                        liblgmi_io.so (csrc/bamio.cpp, include/lgmi_io.h): streaming BGZF reader with
                                        BAI random access — a region query inflates only the blocks its index chunks
                                        point at, so memory follows the region, not the file.  Without a .bai the
                                        index is built in memory by one pass; BamReader.build_index() writes a .bai.
    FastaReader ~ pysam.FastaFile       fetch(contig, start, end); random access through a .fai (built in memory
                                        from one pass over the file when absent), uncompressed FASTA
    VcfReader   ~ pysam.VariantFile     fetch(contig, start, end) -> records with .start (0-based)
    BamWriter                           BGZF writer, used to make synthetic BAMs for tests / benchmarks

Pile-up semantics follow pysam's defaults as far as this path depends on them: reads that are unmapped,
secondary, QC-failed or duplicates are skipped (stepper 'samtools'), orphans of paired reads are skipped,
bases below quality 13 are dropped, at most 8000 reads per column, columns are NOT truncated to the
requested interval, and a read contributes an empty string where it has a deletion or a reference skip.
"""
from __future__ import annotations

import ctypes as C
import bisect
import gzip
import os
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

_SEQ = '=ACMGRSVTWYHKDBN'
_CIGAR = 'MIDNSHP=X'
_BAM_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')

_HERE = os.path.dirname(os.path.abspath(__file__))
IO_LIB_PATH = os.environ.get('LGMI_IO_LIB', os.path.join(os.path.dirname(_HERE), 'lib', 'liblgmi_io.so'))
LGIO_NAMES, LGIO_CIGAR, LGIO_SEQ, LGIO_CS, LGIO_AUX, LGIO_ALL = 1, 2, 4, 8, 16, 31

_u8p, _u16p, _u32p, _u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
_i32p, _i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class _Reads(C.Structure):       # include/lgmi_io.h: lgio_reads
    _fields_ = [('n', C.c_uint64), ('tid', _i32p), ('start', _i64p), ('end', _i64p), ('flag', _u16p), ('mapq', _u8p),
                ('name_off', _u64p), ('names', C.c_void_p), ('cigar_off', _u64p), ('cigar', _u32p),
                ('seq_off', _u64p), ('seq', C.c_void_p), ('qual', _u8p), ('cs_off', _u64p), ('cs', C.c_void_p),
                ('has_cs', _u8p), ('aux_off', _u64p), ('aux', _u8p), ('owner_', C.c_void_p)]


class _Pileup(C.Structure):      # lgio_pileup
    _fields_ = [('n_cols', C.c_uint64), ('pos', _i64p), ('col_off', _u64p), ('read', _u32p), ('base', C.c_void_p),
                ('reads', _Reads), ('owner_', C.c_void_p)]


class SiteParams(C.Structure):   # lgio_site_params
    _fields_ = [('keep_non_spliced_read', C.c_int32), ('min_base_quality', C.c_int32), ('max_depth', C.c_int32),
                ('reserved', C.c_int32), ('min_dist_from_splice', C.c_int64), ('half_window', C.c_int64),
                ('min_allele_depth', C.c_double), ('min_allele_ratio', C.c_double), ('min_total_depth', C.c_double),
                ('max_window_mismatch', C.c_double), ('max_window_mismatch_type', C.c_double)]


class _Sites(C.Structure):       # lgio_sites
    _fields_ = [('fallback', C.c_int32), ('reserved', C.c_int32), ('n_sites', C.c_uint64), ('strand', _u8p), ('pos', _i64p),
                ('ref', C.c_void_p), ('neighbor', _u32p), ('allele_off', _u64p), ('allele_nt', C.c_void_p),
                ('reads_off', _u64p), ('reads', _u32p), ('n_removed', C.c_uint64 * 2), ('removed_pos', _i64p * 2),
                ('removed_code', _u8p * 2), ('n_reads', C.c_uint64), ('name_off', _u64p), ('names', C.c_void_p),
                ('owner_', C.c_void_p), ('read_uid', _u32p)]


class _Intervals(C.Structure):   # lgio_intervals
    _fields_ = [('n', C.c_uint64), ('start', _i64p), ('end', _i64p), ('owner_', C.c_void_p)]


REMOVED_REASONS = ('too many window mismatches', 'too few usable reads after filters', 'not enough allele after filters')

IO_SYMBOLS = {
    'lgio_abi_version': (C.c_int, []),
    'lgio_last_error': (C.c_char_p, []),
    'lgio_bam_open': (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    'lgio_bam_close': (None, [C.c_void_p]),
    'lgio_bam_n_refs': (C.c_int, [C.c_void_p]),
    'lgio_bam_ref_name': (C.c_char_p, [C.c_void_p, C.c_int]),
    'lgio_bam_ref_length': (C.c_int64, [C.c_void_p, C.c_int]),
    'lgio_bam_header_text': (C.c_char_p, [C.c_void_p]),
    'lgio_bam_has_index_file': (C.c_int, [C.c_void_p]),
    'lgio_bam_build_index': (C.c_int, [C.c_char_p, C.c_char_p]),
    'lgio_bam_fetch': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_uint32, C.POINTER(_Reads)]),
    'lgio_reads_free': (None, [C.POINTER(_Reads)]),
    'lgio_bam_pileup': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(_Pileup)]),
    'lgio_pileup_free': (None, [C.POINTER(_Pileup)]),
    'lgio_bam_bytes_read': (C.c_uint64, [C.c_void_p]),
    'lgio_bam_region_sites': (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(SiteParams), C.POINTER(_Sites)]),
    'lgio_sites_free': (None, [C.POINTER(_Sites)]),
    'lgio_bam_ref_intervals': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(_Intervals)]),
    'lgio_intervals_free': (None, [C.POINTER(_Intervals)]),
    'lgio_write_removed_table': (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int8), _i64p,
                                           C.POINTER(C.c_int8), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, C.c_int]),
    'lgio_write_table': (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_void_p, C.c_int]),
    'lgio_format_doubles': (C.c_int, [C.c_uint64, C.POINTER(C.c_double), C.c_char_p, C.c_uint32]),
}


class _TableCol(C.Structure):           # include/lgmi_io.h: lgio_table_col
    _fields_ = [('name', C.c_char_p), ('kind', C.c_uint32), ('n_names', C.c_uint32), ('data', C.c_void_p), ('names', C.POINTER(C.c_char_p))]


def write_table(path, columns, header=True, append=False, threads=4):
    """lgio_write_table: a tab-separated table as pandas' to_csv(sep='\\t', index=False) writes it.  ``columns``: a list of
    (name, values) — an integer array, a float array — or (name, codes, names) for a string column held as dictionary codes.
    Raises ValueError when a name would need quoting (the caller takes pandas)."""
    lib = load_io()
    cols = (_TableCol * max(len(columns), 1))()
    keep, n = [], None
    for k, col in enumerate(columns):
        if len(col) == 3:
            name, codes, names = col
            a = np.ascontiguousarray(codes, np.int32)
            arr = (C.c_char_p * max(len(names), 1))(*[str(x).encode() for x in names])
            cols[k].kind, cols[k].n_names, cols[k].names = 2, len(names), arr
            keep.append(arr)
        else:
            name, values = col
            values = np.asarray(values)
            if values.dtype.kind == 'f':
                a, cols[k].kind = np.ascontiguousarray(values, np.float64), 1
            elif values.dtype.kind in 'iu':
                a, cols[k].kind = np.ascontiguousarray(values, np.int64), 0
            else:
                raise ValueError('column %r: dtype %s is not one the native writer takes' % (name, values.dtype))
        if n is None:
            n = len(a)
        elif len(a) != n:
            raise ValueError('column %r has %d rows, the first one %d' % (name, len(a), n))
        cols[k].name, cols[k].data = str(name).encode(), a.ctypes.data if a.size else None
        keep.append(a)
    _check(lib.lgio_write_table(str(path).encode(), 1 if append else 0, 1 if header else 0, n or 0, len(columns), C.addressof(cols), int(threads)))


def format_doubles(values):
    """lgio_format_doubles: every value as the native table writer writes it (tests: against numpy's astype(str))"""
    lib = load_io()
    a = np.ascontiguousarray(values, np.float64)
    buf = C.create_string_buffer(max(len(a), 1) * 40)
    _check(lib.lgio_format_doubles(len(a), a.ctypes.data_as(C.POINTER(C.c_double)), buf, 40))
    raw = np.frombuffer(buf, dtype='S40', count=len(a))
    return [x.decode() for x in raw.tolist()]


def write_removed_table(path, chrom_names, reason_names, chrom_code, strand, pos, reason_code, header=True, append=False, threads=4):
    """lgio_write_removed_table: the removed-site table of script/giremi.py:403-409 from dictionary codes, the bytes
    pandas' to_csv(sep='\\t', index=False) writes.  Raises ValueError when a name would need quoting (the caller takes pandas)"""
    lib = load_io()
    cc = np.ascontiguousarray(chrom_code, np.int32)
    st = np.ascontiguousarray(strand, np.int8)
    ps = np.ascontiguousarray(pos, np.int64)
    rc_ = np.ascontiguousarray(reason_code, np.int8)
    cn = (C.c_char_p * max(len(chrom_names), 1))(*[str(c).encode() for c in chrom_names])
    rn = (C.c_char_p * max(len(reason_names), 1))(*[str(r).encode() for r in reason_names])
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if a.size else C.cast(None, C.POINTER(t))
    _check(lib.lgio_write_removed_table(str(path).encode(), 1 if append else 0, 1 if header else 0, len(ps), p(cc, C.c_int32), p(st, C.c_int8),
                                        p(ps, C.c_int64), p(rc_, C.c_int8), cn, len(chrom_names), rn, len(reason_names), int(threads)))
_iolib = None


def load_io():
    global _iolib
    if _iolib is None:
        if not os.path.exists(IO_LIB_PATH):
            raise RuntimeError('liblgmi_io.so not found at %s — build it with `make -C l-giremi_amd`' % IO_LIB_PATH)
        lib = C.CDLL(IO_LIB_PATH)
        for name, (res, args) in IO_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.lgio_abi_version() != 4:
            raise RuntimeError('liblgmi_io.so ABI %d != binding ABI 4 (rebuild with `make -C l-giremi_amd`)' % lib.lgio_abi_version())
        _iolib = lib
    return _iolib


def _check(rc):
    if rc:
        msg = load_io().lgio_last_error().decode('utf-8', 'replace')
        raise (OSError if rc == -2 else ValueError)('liblgmi_io error %d: %s' % (rc, msg))


def _arr(ptr, n, dt):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)


def _pool(addr, n):
    return C.string_at(addr, n) if n and addr else b''


class _ReadTable:
    """numpy / bytes copies of an lgio_reads (the library's buffers are released right after the copy)"""

    def __init__(self, rs: _Reads, references):
        n = int(rs.n)
        self.n, self.references = n, references
        self.tid, self.start, self.end = _arr(rs.tid, n, np.int32), _arr(rs.start, n, np.int64), _arr(rs.end, n, np.int64)
        self.flag, self.mapq = _arr(rs.flag, n, np.uint16), _arr(rs.mapq, n, np.uint8)
        self.name_off = _arr(rs.name_off, n + 1, np.int64)
        self.names = _pool(rs.names, int(self.name_off[-1]))
        self.cigar_off = _arr(rs.cigar_off, n + 1, np.int64)
        self.cigar = _arr(rs.cigar, int(self.cigar_off[-1]), np.uint32)
        self.seq_off = _arr(rs.seq_off, n + 1, np.int64)
        self.seq = _pool(rs.seq, int(self.seq_off[-1]))
        self.qual = bytes(_arr(rs.qual, int(self.seq_off[-1]) if rs.qual else 0, np.uint8))
        self.cs_off = _arr(rs.cs_off, n + 1, np.int64)
        self.cs = _pool(rs.cs, int(self.cs_off[-1]))
        self.has_cs = _arr(rs.has_cs, n, np.uint8)
        self.aux_off = _arr(rs.aux_off, n + 1, np.int64)
        self.aux = bytes(_arr(rs.aux, int(self.aux_off[-1]), np.uint8))

    def name(self, k):
        return self.names[self.name_off[k]:self.name_off[k + 1]].decode()


class BamRead:
    """one alignment of a fetched table, with pysam's attribute names"""
    __slots__ = ('_t', '_k', '_tags')

    def __init__(self, table: _ReadTable, k: int):
        self._t, self._k, self._tags = table, k, None

    query_name = property(lambda self: self._t.name(self._k))
    flag = property(lambda self: int(self._t.flag[self._k]))
    reference_id = property(lambda self: int(self._t.tid[self._k]))
    reference_name = property(lambda self: self._t.references[int(self._t.tid[self._k])])
    reference_start = property(lambda self: int(self._t.start[self._k]))
    reference_end = property(lambda self: int(self._t.end[self._k]))
    mapping_quality = property(lambda self: int(self._t.mapq[self._k]))
    is_reverse = property(lambda self: bool(self._t.flag[self._k] & 16))
    is_unmapped = property(lambda self: bool(self._t.flag[self._k] & 4))

    @property
    def cigartuples(self):
        c = self._t.cigar[self._t.cigar_off[self._k]:self._t.cigar_off[self._k + 1]]
        return [(int(v & 0xF), int(v >> 4)) for v in c]

    @property
    def cigarstring(self):
        return ''.join('%d%s' % (n, _CIGAR[op]) for op, n in self.cigartuples)

    @property
    def query_sequence(self):
        return self._t.seq[self._t.seq_off[self._k]:self._t.seq_off[self._k + 1]].decode()

    @property
    def query_qualities(self):
        return self._t.qual[self._t.seq_off[self._k]:self._t.seq_off[self._k + 1]]

    @property
    def tags(self):
        if self._tags is None:
            self._tags = _parse_tags(self._t.aux[self._t.aux_off[self._k]:self._t.aux_off[self._k + 1]])
            if self._t.has_cs[self._k] and 'cs' not in self._tags:
                self._tags['cs'] = self._t.cs[self._t.cs_off[self._k]:self._t.cs_off[self._k + 1]].decode()
        return self._tags

    def get_tag(self, name):
        if name == 'cs' and self._t.has_cs[self._k]:         # the hot one (mismatch.py:76): no aux parsing
            return self._t.cs[self._t.cs_off[self._k]:self._t.cs_off[self._k + 1]].decode()
        return self.tags[name]                                # KeyError like pysam

    def has_tag(self, name):
        return (name == 'cs' and bool(self._t.has_cs[self._k])) or name in self.tags

    def aligned_pairs(self) -> Iterator[Tuple[int, Optional[int]]]:
        """(reference position, query index or None for deletion / reference skip) over the reference span"""
        ref, qry = self.reference_start, 0
        for op, n in self.cigartuples:
            if op in (0, 7, 8):            # M = X
                for k in range(n):
                    yield ref + k, qry + k
                ref += n
                qry += n
            elif op in (2, 3):             # D N
                for k in range(n):
                    yield ref + k, None
                ref += n
            elif op in (1, 4):             # I S
                qry += n


def _parse_tags(buf: bytes) -> Dict[str, object]:
    tags, i, n = {}, 0, len(buf)
    fmt = {'c': 'b', 'C': 'B', 's': 'h', 'S': 'H', 'i': 'i', 'I': 'I', 'f': 'f'}
    while i + 3 <= n:
        name, typ = buf[i:i + 2].decode(), chr(buf[i + 2])
        i += 3
        if typ == 'A':
            tags[name] = chr(buf[i]); i += 1
        elif typ in fmt:
            size = struct.calcsize(fmt[typ])
            tags[name] = struct.unpack_from('<' + fmt[typ], buf, i)[0]; i += size
        elif typ in 'ZH':
            j = buf.index(b'\0', i)
            tags[name] = buf[i:j].decode(); i = j + 1
        elif typ == 'B':
            sub = chr(buf[i]); cnt = struct.unpack_from('<i', buf, i + 1)[0]; i += 5
            size = struct.calcsize(fmt[sub])
            tags[name] = list(struct.unpack_from('<%d%s' % (cnt, fmt[sub]), buf, i)); i += cnt * size
        else:
            raise ValueError('unknown BAM tag type %r' % typ)
    return tags


class BamReader:
    """pysam.AlignmentFile stand-in on liblgmi_io.so: indexed, streaming (see the module docstring)"""

    def __init__(self, path: str):
        self._lib = load_io()
        h = C.c_void_p()
        _check(self._lib.lgio_bam_open(os.fsencode(path), C.byref(h)))
        self._h, self.path = h, path
        n = self._lib.lgio_bam_n_refs(h)
        self.references = [self._lib.lgio_bam_ref_name(h, k).decode() for k in range(n)]
        self.lengths = [int(self._lib.lgio_bam_ref_length(h, k)) for k in range(n)]
        self.header_text = (self._lib.lgio_bam_header_text(h) or b'').decode(errors='replace')
        self._tid = {r: k for k, r in enumerate(self.references)}

    @staticmethod
    def build_index(path: str, bai_path: Optional[str] = None):
        """one streaming pass -> standard .bai next to the BAM"""
        _check(load_io().lgio_bam_build_index(os.fsencode(path), os.fsencode(bai_path) if bai_path else None))

    @property
    def has_index_file(self):
        return bool(self._lib.lgio_bam_has_index_file(self._h))

    @property
    def bytes_read(self):
        return int(self._lib.lgio_bam_bytes_read(self._h))

    def close(self):
        if self._h:
            self._lib.lgio_bam_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _table(self, contig, start, stop, what) -> Optional[_ReadTable]:
        tid = self._tid.get(contig)
        if tid is None:
            return None
        rs = _Reads()
        _check(self._lib.lgio_bam_fetch(self._h, tid, -1 if start is None else int(start), 0 if stop is None else int(stop),
                                        what, C.byref(rs)))
        try:
            return _ReadTable(rs, self.references)
        finally:
            self._lib.lgio_reads_free(C.byref(rs))

    def intervals(self, contig, threads=None):
        """(starts, ends) of the mapped reads of a contig as numpy arrays — what the footprint merge needs, without
        materialising names, sequences or tags (src/giremi/footprint.py:6-28).  The scan reads the whole contig: its BGZF
        blocks are inflated by `threads` threads (default: the CPUs this process may use, at most 16)"""
        tid = self._tid.get(contig)
        if tid is None:
            raise KeyError(contig)
        if threads is None:
            threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
        iv = _Intervals()
        _check(self._lib.lgio_bam_ref_intervals(self._h, tid, int(threads), C.byref(iv)))
        try:
            n = int(iv.n)
            return _arr(iv.start, n, np.int64), _arr(iv.end, n, np.int64)
        finally:
            self._lib.lgio_intervals_free(C.byref(iv))

    def fetch(self, contig=None, start=None, stop=None):
        t = self._table(contig, start, stop, LGIO_ALL)
        for k in range(t.n if t is not None else 0):
            yield BamRead(t, k)

    def region_sites(self, contig, start, stop, params: SiteParams):
        """lgio_bam_region_sites: the BAM-only steps of the site extraction of one footprint (lgmi.region uses it).
        -> None when the library leaves the footprint to the Python path, else a dict of arrays:
        strand/pos/ref/neighbor/allele_off/allele_nt/reads_off/reads (surviving sites), removed = [(pos, code)] per
        strand (codes index REMOVED_REASONS), names = the read names the read ids refer to"""
        tid = self._tid.get(contig)
        if tid is None:
            return None
        st = _Sites()
        _check(self._lib.lgio_bam_region_sites(self._h, tid, int(start), int(stop), C.byref(params), C.byref(st)))
        try:
            if st.fallback:
                return None
            n = int(st.n_sites)
            aoff = _arr(st.allele_off, n + 1, np.int64)
            na = int(aoff[-1]) if n else 0
            roff = _arr(st.reads_off, na + 1, np.int64)
            nr = int(st.n_reads)
            noff = _arr(st.name_off, nr + 1, np.int64)
            return {'strand': _arr(st.strand, n, np.uint8), 'pos': _arr(st.pos, n, np.int64), 'ref': _pool(st.ref, n),
                    'neighbor': _arr(st.neighbor, 16 * n, np.int64).reshape(n, 16), 'allele_off': aoff,
                    'allele_nt': _pool(st.allele_nt, na), 'reads_off': roff,
                    'reads': _arr(st.reads, int(roff[-1]) if na else 0, np.int64),
                    'removed': [(_arr(st.removed_pos[k], int(st.n_removed[k]), np.int64),
                                 _arr(st.removed_code[k], int(st.n_removed[k]), np.uint8)) for k in range(2)],
                    'name_off': noff, 'names': _pool(st.names, int(noff[-1]) if nr else 0), 'read_uid': _arr(st.read_uid, nr, np.int64)}
        finally:
            self._lib.lgio_sites_free(C.byref(st))

    def pileup(self, contig=None, start=None, stop=None, min_base_quality=13, max_depth=8000):
        """columns as pysam's pileup(contig, start, stop) yields them for single-end (long) reads.  Paired short reads
        would differ: overlapping mates are both emitted (htslib drops one), deletion / reference-skip entries bypass
        min_base_quality, max_depth caps columns rather than incoming reads — include/lgmi_io.h lists the three."""
        tid = self._tid.get(contig)
        if tid is None:
            return
        pl = _Pileup()
        _check(self._lib.lgio_bam_pileup(self._h, tid, -1 if start is None else int(start), 0 if stop is None else int(stop),
                                         int(min_base_quality), int(max_depth), C.byref(pl)))
        try:
            nc = int(pl.n_cols)
            pos, off = _arr(pl.pos, nc, np.int64), _arr(pl.col_off, nc + 1, np.int64)
            ridx = _arr(pl.read, int(off[-1]) if nc else 0, np.int64)
            base = _pool(pl.base, int(off[-1]) if nc else 0)
            nr = int(pl.reads.n)
            noff = _arr(pl.reads.name_off, nr + 1, np.int64)
            pool = _pool(pl.reads.names, int(noff[-1]) if nr else 0)
            names = [pool[noff[k]:noff[k + 1]].decode() for k in range(nr)]
        finally:
            self._lib.lgio_pileup_free(C.byref(pl))
        # columns are views: the caller looks at the names / bases of the few columns that are candidate sites
        # (mismatch.py:160-190) and only at the position of all the others
        for c in range(nc):
            yield PileupColumn(int(pos[c]), names, ridx, base, int(off[c]), int(off[c + 1]))


class PileupColumn:
    __slots__ = ('pos', 'reference_pos', '_names', '_ridx', '_base', '_a', '_b')

    def __init__(self, pos, names, ridx=None, base=None, a=0, b=0):
        self.pos = self.reference_pos = pos
        self._names, self._ridx, self._base, self._a, self._b = names, ridx, base, a, b

    def get_query_names(self):
        names = self._names
        return [names[r] for r in self._ridx[self._a:self._b].tolist()]

    def get_query_sequences(self):
        return [chr(x) if x else '' for x in self._base[self._a:self._b]]


def _bgzf_block(payload: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', bsize) + body +
            struct.pack('<II', zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF))


class BamWriter:
    """writes coordinate-sorted, single-end, mapped records with a cs:Z tag (what minimap2 --cs emits)"""

    def __init__(self, path: str, references: List[Tuple[str, int]], index: bool = True):
        self.path, self.references, self.index = path, list(references), index
        self._ids = {name: k for k, (name, _l) in enumerate(self.references)}
        text = '@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in self.references)
        self._buf = bytearray(b'BAM\1' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(self.references)))
        for name, ln in self.references:
            self._buf += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', ln)

    def write(self, contig: str, start: int, name: str, is_reverse: bool, cigartuples, sequence: str, cs: str,
              mapq: int = 60, quality=40, flag: int = 0, cs_tag: bool = True):
        """quality: one phred value for every base, or a sequence of per-base values; flag: extra SAM flag bits;
        cs_tag=False writes the record without a cs tag (an NM:i tag instead)"""
        code = {c: k for k, c in enumerate(_SEQ)}
        seq = sequence.upper()
        packed = bytearray()
        for k in range(0, len(seq), 2):
            hi = code.get(seq[k], 15)
            lo = code.get(seq[k + 1], 15) if k + 1 < len(seq) else 0
            packed.append((hi << 4) | lo)
        cig = b''.join(struct.pack('<I', (n << 4) | op) for op, n in cigartuples)
        tags = b'csZ' + cs.encode() + b'\0' if cs_tag else b'NMC\x00'
        end = start + sum(n for op, n in cigartuples if op in (0, 2, 3, 7, 8))
        core = struct.pack('<iiBBHHHiiii', self._ids[contig], start, len(name) + 1, mapq, _reg2bin(start, end),
                           len(cigartuples), (16 if is_reverse else 0) | flag, len(seq), -1, -1, 0)
        quals = bytes([quality]) * len(seq) if isinstance(quality, int) else bytes(quality)
        rec = core + name.encode() + b'\0' + cig + bytes(packed) + quals + tags
        self._buf += struct.pack('<i', len(rec)) + rec

    def close(self):
        with open(self.path, 'wb') as f:
            data = bytes(self._buf)
            for k in range(0, len(data), 60000):
                f.write(_bgzf_block(data[k:k + 60000]))
            f.write(_BAM_EOF)
        if self.index:
            BamReader.build_index(self.path)


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


# ---------------------------------------------------------------------------------------------- FASTA / VCF / repeats
class FastaReader:
    """pysam.FastaFile stand-in: random access through the .fai (name, length, offset, bases per line, bytes per
    line); the index is built in memory by one pass over the file when there is no .fai next to it.  gzip input
    is read whole (no random access into plain gzip)."""

    def __init__(self, path: str, fai=None):
        """``fai``: the index another FastaReader of the same file built (FastaReader.index): no .fai is read, no pass made"""
        self.path = path
        self._fai: Dict[str, Tuple[int, int, int, int]] = {}
        self._seq: Optional[Dict[str, str]] = None
        if fai is not None and not path.endswith('.gz'):
            self._fai = dict(fai)
            self.references = list(self._fai)
            self._f = open(path, 'rb')
            return
        if path.endswith('.gz'):
            self._seq = {}
            name, parts = None, []
            with gzip.open(path, 'rt') as f:
                for line in f:
                    if line.startswith('>'):
                        if name is not None:
                            self._seq[name] = ''.join(parts)
                        name, parts = line[1:].split()[0], []
                    else:
                        parts.append(line.strip())
            if name is not None:
                self._seq[name] = ''.join(parts)
            self.references = list(self._seq)
            self._f = None
            return
        if os.path.exists(path + '.fai'):
            with open(path + '.fai') as f:
                for line in f:
                    c = line.rstrip('\n').split('\t')
                    if len(c) >= 5:
                        self._fai[c[0]] = (int(c[1]), int(c[2]), int(c[3]), int(c[4]))
        else:
            self._fai = self._scan_fast(path)
        self.references = list(self._fai)
        self._f = open(path, 'rb')

    @property
    def index(self):
        """name -> (length, offset, bases per line, bytes per line), or None for a gzip file held in memory"""
        return None if self._seq is not None else dict(self._fai)

    @classmethod
    def _scan_fast(cls, path):
        """_scan without a Python step per line: the header lines are found with mmap.find, a sequence's length is its bytes
        minus its line ends (bytes.count over pieces of the map).  (_scan reads the 800,000 lines of a 49-MB genome one by
        one: 0.3 s at the start of every run on a FASTA without a .fai.)  Falls back to _scan for what it does not cover: a
        sequence whose first line is empty."""
        import mmap
        size = os.path.getsize(path)
        if size == 0:
            return {}
        piece = 1 << 26

        def count(mm, a, b, ch):
            return sum(mm[o:min(b, o + piece)].count(ch) for o in range(a, b, piece))

        with open(path, 'rb') as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            heads = [0] if mm[0:1] == b'>' else []                               # '>' at the start of a line
            at = mm.find(b'\n>')
            while at >= 0:
                heads.append(at + 1)
                at = mm.find(b'\n>', at + 1)
            fai = {}
            for k, h in enumerate(heads):
                end = heads[k + 1] if k + 1 < len(heads) else size
                nl = mm.find(b'\n', h, end)
                head_end = end if nl < 0 else nl + 1                             # (a header that is the file's last line, unterminated)
                name = mm[h + 1:head_end].split()[0].decode()
                n_body = end - head_end
                n_base = n_body - count(mm, head_end, end, b'\n') - count(mm, head_end, end, b'\r')
                lb = lw = 0
                if n_body:
                    first = mm.find(b'\n', head_end, end)
                    line = mm[head_end:end if first < 0 else first + 1]
                    lb, lw = len(line.rstrip(b'\r\n')), len(line)
                    if lb == 0 and n_base > 0:
                        return cls._scan(path)                                    # bases after an empty first line: the line-by-line pass
                    if lb == 0:
                        lw = 0
                # (_scan strips CR / LF at line ends only; a CR inside a line is not FASTA)
                fai[name] = (n_base, head_end, lb, lw)
            return fai

    @staticmethod
    def _scan(path):
        fai, name, length, offset, lb, lw, at = {}, None, 0, 0, 0, 0, 0
        with open(path, 'rb') as f:
            for line in f:
                if line.startswith(b'>'):
                    if name is not None:
                        fai[name] = (length, offset, lb, lw)
                    name, length, offset, lb, lw = line[1:].split()[0].decode(), 0, at + len(line), 0, 0
                else:
                    body = line.rstrip(b'\r\n')
                    if lb == 0 and body:
                        lb, lw = len(body), len(line)
                    length += len(body)
                at += len(line)
        if name is not None:
            fai[name] = (length, offset, lb, lw)
        return fai

    def fetch(self, contig, start=None, end=None):
        if self._seq is not None:
            s = self._seq[contig]
            return s if start is None else s[max(start, 0):end]
        length, offset, lb, lw = self._fai[contig]
        a = 0 if start is None else max(int(start), 0)
        b = length if end is None else min(int(end), length)
        if a >= b or lb == 0:
            return ''
        first = offset + (a // lb) * lw + a % lb
        last = offset + ((b - 1) // lb) * lw + (b - 1) % lb + 1
        self._f.seek(first)
        return self._f.read(last - first).replace(b'\n', b'').replace(b'\r', b'').decode()

    def close(self):
        if getattr(self, '_f', None):
            self._f.close()
            self._f = None


class _VcfRecord:
    __slots__ = ('contig', 'start', 'stop')

    def __init__(self, contig, start, stop):
        self.contig, self.start, self.stop = contig, start, stop


class VcfReader:
    """text VCF (optionally gzip/bgzip); records overlapping [start, end) are returned with 0-based .start"""

    def __init__(self, path: str):
        self._rec: Dict[str, List[_VcfRecord]] = {}
        with open(path, 'rb') as probe:
            magic = probe.read(2)
        opener = gzip.open if magic == b'\x1f\x8b' else open
        with opener(path, 'rt') as f:
            for line in f:
                if line.startswith('#') or not line.strip():
                    continue
                c = line.split('\t')
                start = int(c[1]) - 1
                self._rec.setdefault(c[0], []).append(_VcfRecord(c[0], start, start + max(len(c[3]), 1)))
        self._starts: Dict[str, List[int]] = {}
        self._longest: Dict[str, int] = {}
        for c, v in self._rec.items():
            v.sort(key=lambda r: r.start)
            self._starts[c] = [r.start for r in v]
            self._longest[c] = max(r.stop - r.start for r in v)

    def fetch(self, contig, start=None, end=None):
        recs = self._rec.get(contig, [])
        if start is None:
            yield from recs
            return
        # records are sorted by start: those that can overlap [start, end) begin before `end` and no more than the
        # longest record before `start` (a scan of the whole contig per query was quadratic over a run's footprints)
        starts = self._starts.get(contig, [])
        lo = bisect.bisect_left(starts, start - self._longest.get(contig, 1) + 1)
        hi = bisect.bisect_left(starts, end)
        for r in recs[lo:hi]:
            if r.stop > start:
                yield r

    def close(self):
        pass


def open_alignment(path):
    return BamReader(path)


def open_fasta(path, fai=None):
    return FastaReader(path, fai=fai)


def open_variants(path):
    return VcfReader(path)
