"""ctypes loader of the C oracle (oracle/lgmi_oracle.c -> oracle/_build/liblgmi_oracle.so).

TEST INFRASTRUCTURE ONLY — see the header of lgmi_oracle.c.  Builds the shared
object on first use with the committed Makefile (gcc only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'liblgmi_oracle.so')

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
i64p, f64p = C.POINTER(C.c_int64), C.POINTER(C.c_double)


class CBatch(C.Structure):  # include/lgmi.h: lgmi_batch
    _fields_ = [('n_blocks', C.c_uint64), ('n_sites', C.c_uint64), ('n_plane_words', C.c_uint64),
                ('block_site_begin', u64p), ('block_n_reads', u32p), ('site_pos', i64p),
                ('site_type', u8p), ('site_word_off', u32p), ('site_n_words', u32p),
                ('site_plane_off', u64p), ('planes', u64p)]


class CResult(C.Structure):  # lgmi_oracle.c: lgo_result
    _fields_ = [('n_rows', C.c_uint64), ('n_examined', C.c_uint64), ('row_i', u32p), ('row_j', u32p),
                ('row_mi', f64p), ('row_counts', u32p), ('row_p', f64p), ('row_exceed', u32p),
                ('site_mean_mi', f64p), ('site_n_pairs', u32p)]


_lib = None


def build(force=False):
    src_newer = (not os.path.exists(SO)) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(SO)
        for f in ('lgmi_oracle.c', 'lgmi_perm_oracle.c', 'Makefile'))
    if force or src_newer:
        subprocess.run(['make', '-C', HERE, '-B' if force else '-s'], check=True, capture_output=True)
    return SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(SO)
        lib.lgo_run.restype = C.c_int
        lib.lgo_run.argtypes = [C.POINTER(CBatch), C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_int,
                                C.POINTER(CResult)]
        lib.lgo_free.argtypes = [C.POINTER(CResult)]
        lib.lgo_mi_from_table.restype = C.c_double
        lib.lgo_mi_from_table.argtypes = [u32p]
        lib.lgo_num_threads.restype = C.c_int
        lib.lgo_perm_rows.restype = C.c_int
        lib.lgo_perm_rows.argtypes = [C.c_uint64, u32p, u32p, u32p, C.c_uint32, C.c_uint64, f64p, u32p, C.c_int]
        lib.lgo_perm_rows_exact.restype = C.c_int
        lib.lgo_perm_rows_exact.argtypes = [C.c_uint64, u32p, f64p]
        _lib = lib
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(t) if a.size else C.cast(None, t)


def batch_struct(pb):
    """pb: any object with the numpy fields of lgmi.pack.PackedBatch"""
    return CBatch(len(pb.block_n_reads), len(pb.site_pos), pb.planes.size,
                  _ptr(pb.block_site_begin, u64p), _ptr(pb.block_n_reads, u32p), _ptr(pb.site_pos, i64p),
                  _ptr(pb.site_type, u8p), _ptr(pb.site_word_off, u32p), _ptr(pb.site_n_words, u32p),
                  _ptr(pb.site_plane_off, u64p), _ptr(pb.planes, u64p))


def run(pb, min_common=5, het_only=True, n_shuffles=0, seed=0, threads=0):
    """-> dict(row_i,row_j,row_mi,row_counts[,row_p,row_exceed],site_mean_mi,site_n_pairs,n_examined)"""
    lib = load()
    st, res = batch_struct(pb), CResult()
    rc = lib.lgo_run(C.byref(st), int(min_common), 1 if het_only else 0, int(n_shuffles), int(seed) & (2**64 - 1),
                     int(threads), C.byref(res))
    if rc == -7:
        raise ValueError('math domain error')
    if rc:
        raise RuntimeError('lgo_run failed: %d' % rc)
    try:
        n, ns = int(res.n_rows), len(pb.site_pos)

        def a(ptr, count, dt):
            if not ptr:
                return None
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dt, copy=True) if count else np.zeros(0, dt)
        out = {'row_i': a(res.row_i, n, np.uint32), 'row_j': a(res.row_j, n, np.uint32),
               'row_mi': a(res.row_mi, n, np.float64),
               'row_counts': a(res.row_counts, 9 * n, np.uint32).reshape(n, 3, 3),
               'row_p': a(res.row_p, n, np.float64), 'row_exceed': a(res.row_exceed, n, np.uint32),
               'site_mean_mi': a(res.site_mean_mi, ns, np.float64),
               'site_n_pairs': a(res.site_n_pairs, ns, np.uint32), 'n_examined': int(res.n_examined)}
        return out
    finally:
        lib.lgo_free(C.byref(res))


def mi_from_table(table9):
    t = np.ascontiguousarray(table9, np.uint32).reshape(9)
    return float(load().lgo_mi_from_table(t.ctypes.data_as(u32p)))


def perm_rows(row_i, row_j, counts, n_shuffles, seed, threads=0):
    """the permutation specification on caller-supplied tables -> (p, exceed)"""
    lib = load()
    ri = np.ascontiguousarray(row_i, np.uint32)
    rj = np.ascontiguousarray(row_j, np.uint32)
    c = np.ascontiguousarray(counts, np.uint32).reshape(-1, 9)
    p = np.empty(len(ri), np.float64)
    e = np.empty(len(ri), np.uint32)
    rc = lib.lgo_perm_rows(len(ri), _ptr(ri, u32p), _ptr(rj, u32p), _ptr(c, u32p), int(n_shuffles),
                           int(seed) & (2**64 - 1), _ptr(p, f64p), _ptr(e, u32p), int(threads))
    if rc:
        raise RuntimeError('lgo_perm_rows failed: %d' % rc)
    return p, e


def perm_rows_exact(counts):
    """exact permutation p of <= 2 x 2 tables (NaN for larger ones): what lgmi_params.exact_2x2 returns"""
    lib = load()
    c = np.ascontiguousarray(counts, np.uint32).reshape(-1, 9)
    p = np.empty(len(c), np.float64)
    rc = lib.lgo_perm_rows_exact(len(c), _ptr(c, u32p), _ptr(p, f64p))
    if rc:
        raise RuntimeError('lgo_perm_rows_exact failed: %d' % rc)
    return p
