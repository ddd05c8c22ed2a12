"""Minimal sequence-file access for the l-giremi-compatible CLI.

The reference opens its inputs with pysam (src/giremi/script/giremi.py:21-24); pysam/htslib are
not installable in this environment, so this module provides small pure-Python stand-ins with the
pysam attributes that the path actually touches (src/giremi/mismatch.py:69-190,
src/giremi/footprint.py:6-28, src/giremi/fileio.py:24-30):

    BamReader   ~ pysam.AlignmentFile   fetch(), pileup(); reads expose query_name, reference_start,
                                        reference_end, is_reverse, get_tag()
    FastaReader ~ pysam.FastaFile       fetch(contig, start, end)
    VcfReader   ~ pysam.VariantFile     fetch(contig, start, end) -> records with .start (0-based)
    BamWriter                           BGZF writer, used to make synthetic BAMs for tests / benchmarks

``open_alignment`` / ``open_fasta`` / ``open_variants`` return pysam objects when pysam is importable
(real data, indexed random access) and these readers otherwise.  The readers load a whole file into
memory: adequate for tests and modest inputs, not for a 100-GB BAM.

Pile-up semantics follow pysam's defaults as far as this path depends on them: reads that are unmapped,
secondary, QC-failed or duplicates are skipped (stepper 'samtools'), orphans of paired reads are skipped,
bases below quality 13 are dropped, at most 8000 reads per column, columns are NOT truncated to the
requested interval, and a read contributes an empty string where it has a deletion or a reference skip.
"""
from __future__ import annotations

import gzip
import struct
import zlib
from typing import Dict, Iterator, List, Optional, Tuple

_SEQ = '=ACMGRSVTWYHKDBN'
_CIGAR = 'MIDNSHP=X'
_BAM_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')


# ---------------------------------------------------------------------------------------------- BAM
class BamRead:
    __slots__ = ('query_name', 'flag', 'reference_id', 'reference_name', 'reference_start', 'mapping_quality',
                 'cigartuples', 'query_sequence', 'query_qualities', 'tags', 'reference_end')

    @property
    def is_reverse(self):
        return bool(self.flag & 16)

    @property
    def is_unmapped(self):
        return bool(self.flag & 4)

    @property
    def cigarstring(self):
        return ''.join('%d%s' % (n, _CIGAR[op]) for op, n in self.cigartuples)

    def get_tag(self, name):
        return self.tags[name]            # KeyError like pysam

    def has_tag(self, name):
        return name in self.tags

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
    def __init__(self, path: str):
        with gzip.open(path, 'rb') as f:      # BGZF is a series of gzip members
            data = f.read()
        if data[:4] != b'BAM\1':
            raise ValueError('%s is not a BAM file' % path)
        l_text = struct.unpack_from('<i', data, 4)[0]
        self.header_text = data[8:8 + l_text].decode(errors='replace')
        at = 8 + l_text
        n_ref = struct.unpack_from('<i', data, at)[0]; at += 4
        self.references, self.lengths = [], []
        for _ in range(n_ref):
            ln = struct.unpack_from('<i', data, at)[0]; at += 4
            self.references.append(data[at:at + ln - 1].decode()); at += ln
            self.lengths.append(struct.unpack_from('<i', data, at)[0]); at += 4
        self._by_ref: Dict[str, List[BamRead]] = {r: [] for r in self.references}
        while at + 4 <= len(data):
            block = struct.unpack_from('<i', data, at)[0]; at += 4
            rec = data[at:at + block]; at += block
            (ref_id, pos, l_name, mapq, _bin, n_cig, flag, l_seq, _nref, _npos, _tlen) = struct.unpack_from('<iiBBHHHiiii', rec, 0)
            r = BamRead()
            p = 32
            r.query_name = rec[p:p + l_name - 1].decode(); p += l_name
            cig = struct.unpack_from('<%dI' % n_cig, rec, p); p += 4 * n_cig
            r.cigartuples = [(c & 0xF, c >> 4) for c in cig]
            nb = (l_seq + 1) // 2
            packed = rec[p:p + nb]; p += nb
            r.query_sequence = ''.join(_SEQ[b >> 4] + _SEQ[b & 0xF] for b in packed)[:l_seq]
            r.query_qualities = rec[p:p + l_seq]; p += l_seq
            r.tags = _parse_tags(rec[p:])
            r.flag, r.reference_id, r.reference_start, r.mapping_quality = flag, ref_id, pos, mapq
            r.reference_name = self.references[ref_id] if ref_id >= 0 else None
            r.reference_end = pos + sum(n for op, n in r.cigartuples if op in (0, 2, 3, 7, 8))
            if ref_id >= 0:
                self._by_ref[r.reference_name].append(r)

    def close(self):
        pass

    def fetch(self, contig=None, start=None, stop=None):
        for r in self._by_ref.get(contig, []):
            if r.is_unmapped:
                continue
            if start is None or (r.reference_end > start and r.reference_start < stop):
                yield r

    def pileup(self, contig=None, start=None, stop=None, min_base_quality=13, max_depth=8000):
        cols: Dict[int, Tuple[List[str], List[str]]] = {}
        for r in self.fetch(contig, start, stop):
            if r.flag & (4 | 256 | 512 | 1024):                    # unmapped, secondary, qc-fail, duplicate
                continue
            if (r.flag & 1) and not (r.flag & 2):                  # orphan of a paired read
                continue
            seq, qual = r.query_sequence, r.query_qualities
            for ref_pos, q in r.aligned_pairs():
                if q is None:
                    base = ''
                else:
                    if qual and qual[q] != 0xFF and qual[q] < min_base_quality:
                        continue
                    base = seq[q]
                names, bases = cols.setdefault(ref_pos, ([], []))
                if len(names) < max_depth:
                    names.append(r.query_name)
                    bases.append(base)
        for pos in sorted(cols):
            yield PileupColumn(pos, *cols[pos])


class PileupColumn:
    def __init__(self, pos, names, bases):
        self.pos, self._names, self._bases = pos, names, bases
        self.reference_pos = pos

    def get_query_names(self):
        return list(self._names)

    def get_query_sequences(self):
        return list(self._bases)


def _bgzf_block(payload: bytes) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = comp.compress(payload) + comp.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    return (b'\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00' + struct.pack('<H', bsize) + body +
            struct.pack('<II', zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF))


class BamWriter:
    """writes coordinate-sorted, single-end, mapped records with a cs:Z tag (what minimap2 --cs emits)"""

    def __init__(self, path: str, references: List[Tuple[str, int]]):
        self.path, self.references = path, list(references)
        self._ids = {name: k for k, (name, _l) in enumerate(self.references)}
        text = '@HD\tVN:1.6\tSO:coordinate\n' + ''.join('@SQ\tSN:%s\tLN:%d\n' % r for r in self.references)
        self._buf = bytearray(b'BAM\1' + struct.pack('<i', len(text)) + text.encode() + struct.pack('<i', len(self.references)))
        for name, ln in self.references:
            self._buf += struct.pack('<i', len(name) + 1) + name.encode() + b'\0' + struct.pack('<i', ln)

    def write(self, contig: str, start: int, name: str, is_reverse: bool, cigartuples, sequence: str, cs: str,
              mapq: int = 60, quality: int = 40):
        code = {c: k for k, c in enumerate(_SEQ)}
        seq = sequence.upper()
        packed = bytearray()
        for k in range(0, len(seq), 2):
            hi = code.get(seq[k], 15)
            lo = code.get(seq[k + 1], 15) if k + 1 < len(seq) else 0
            packed.append((hi << 4) | lo)
        cig = b''.join(struct.pack('<I', (n << 4) | op) for op, n in cigartuples)
        tags = b'csZ' + cs.encode() + b'\0'
        end = start + sum(n for op, n in cigartuples if op in (0, 2, 3, 7, 8))
        core = struct.pack('<iiBBHHHiiii', self._ids[contig], start, len(name) + 1, mapq, _reg2bin(start, end),
                           len(cigartuples), 16 if is_reverse else 0, len(seq), -1, -1, 0)
        rec = core + name.encode() + b'\0' + cig + bytes(packed) + bytes([quality]) * len(seq) + tags
        self._buf += struct.pack('<i', len(rec)) + rec

    def close(self):
        with open(self.path, 'wb') as f:
            data = bytes(self._buf)
            for k in range(0, len(data), 60000):
                f.write(_bgzf_block(data[k:k + 60000]))
            f.write(_BAM_EOF)


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


# ---------------------------------------------------------------------------------------------- FASTA / VCF / repeats
class FastaReader:
    def __init__(self, path: str):
        self._seq: Dict[str, str] = {}
        opener = gzip.open if path.endswith('.gz') else open
        name, parts = None, []
        with opener(path, 'rt') as f:
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

    def fetch(self, contig, start=None, end=None):
        s = self._seq[contig]
        return s if start is None else s[max(start, 0):end]

    def close(self):
        pass


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
        for v in self._rec.values():
            v.sort(key=lambda r: r.start)

    def fetch(self, contig, start=None, end=None):
        for r in self._rec.get(contig, []):
            if start is None or (r.stop > start and r.start < end):
                yield r

    def close(self):
        pass


def open_alignment(path):
    try:
        import pysam
        return pysam.AlignmentFile(path, 'rb')
    except ImportError:
        return BamReader(path)


def open_fasta(path):
    try:
        import pysam
        return pysam.FastaFile(path)
    except ImportError:
        return FastaReader(path)


def open_variants(path):
    try:
        import pysam
        return pysam.VariantFile(path)
    except ImportError:
        return VcfReader(path)
