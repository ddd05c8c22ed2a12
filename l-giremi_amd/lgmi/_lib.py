"""ctypes binding of liblgmi.so (include/lgmi.h) — the only way the Python host
reaches the HIP kernels.  No torch, no fallback: if the shared library is missing
or no MI355X is usable, calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LGMI_LIB', os.path.join(os.path.dirname(_HERE), 'lib', 'liblgmi.so'))

ABI_VERSION = 6
OK, E_ARG, E_OOM, E_HIP, E_RCCL, E_NODEV, E_STATE, E_DOMAIN = 0, -1, -2, -3, -4, -5, -6, -7
TYPE_MISMATCH, TYPE_SNP, TYPE_HET_SNP = 0, 1, 2
UNIQUE_ID_BYTES = 128
EMIT_SEG = 8192
EMIT_SEG_Q = 1024
EXCEED_EXACT = 0xFFFFFFFF
NONE = 0xFFFFFFFF

u8p, u32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
u16p = C.POINTER(C.c_uint16)
i64p, f64p = C.POINTER(C.c_int64), C.POINTER(C.c_double)


class Batch(C.Structure):
    _fields_ = [('n_blocks', C.c_uint64), ('n_sites', C.c_uint64), ('n_plane_words', C.c_uint64),
                ('block_site_begin', u64p), ('block_n_reads', u32p), ('site_pos', i64p),
                ('site_type', u8p), ('site_word_off', u32p), ('site_n_words', u32p),
                ('site_plane_off', u64p), ('planes', u64p), ('site_tri', u8p)]


class Params(C.Structure):
    _fields_ = [('min_common', C.c_uint32), ('n_shuffles', C.c_uint32), ('seed', C.c_uint64),
                ('het_only', C.c_uint8), ('emit_counts', C.c_uint8), ('exact_2x2', C.c_uint8),
                ('no_row_p', C.c_uint8), ('shard_rank', C.c_uint16), ('shard_world', C.c_uint16),
                ('stream_site_base', C.c_uint32), ('compact_rows', C.c_uint8), ('reserved1', C.c_uint8 * 3)]


class Result(C.Structure):
    _fields_ = [('n_rows', C.c_uint64), ('n_sites', C.c_uint64), ('row_i', u32p), ('row_j', u32p),
                ('row_mi', f64p), ('row_p', f64p), ('row_exceed', u32p), ('row_counts', u32p),
                ('site_mean_mi', f64p), ('site_n_pairs', u32p), ('owner_', C.c_void_p),
                ('n_shuffles', C.c_uint32), ('row_p_derived', C.c_uint32),
                # ABI 6: the compact row form (include/lgmi.h)
                ('compact', C.c_uint32), ('reserved3', C.c_uint32), ('row_begin', u64p), ('site_row_full', u8p),
                ('row_j_listed', u32p), ('n_row_j_listed', C.c_uint64), ('row_exceed16', u16p)]


class RunInfo(C.Structure):
    _fields_ = [('n_rows', C.c_uint64), ('n_examined', C.c_uint64), ('n_tile_pairs', C.c_uint64),
                ('word_pairs', C.c_uint64), ('bytes_in', C.c_uint64), ('bytes_out', C.c_uint64),
                ('ms_total', C.c_float), ('ms_prep', C.c_float), ('ms_count', C.c_float),
                ('ms_emit', C.c_float), ('ms_perm', C.c_float), ('ms_mean', C.c_float),
                ('n_count_launches', C.c_uint32), ('n_mfma_tiles', C.c_uint32),
                ('mfma_dtype', C.c_uint32), ('n_six_rows', C.c_uint32),
                ('n_examined_total', C.c_uint64), ('n_general_rows', C.c_uint64),
                ('ms_plan_host', C.c_float), ('ms_perm_fast', C.c_float), ('ms_perm_general', C.c_float),
                ('n_seq_shards', C.c_uint32), ('ms_perm_exact', C.c_float), ('reserved2', C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith('reserved')}


class ShardPlan(C.Structure):
    _fields_ = [('n_items_total', C.c_uint64), ('item_begin', C.c_uint64), ('item_end', C.c_uint64),
                ('n_examined_total', C.c_uint64), ('n_examined', C.c_uint64),
                ('n_tiles_total', C.c_uint64), ('n_tiles', C.c_uint64),
                ('item_site', u32p), ('item_seg', u32p),
                ('tile_block', u32p), ('tile_x0', u32p), ('tile_y0', u32p), ('tile_edge', u32p),
                ('site_xrow', u32p), ('site_ycol', u32p), ('site_prow', u32p), ('site_pcol', u32p),
                ('site_xnext', u32p), ('owner_', C.c_void_p)]


class GatherOpts(C.Structure):
    _fields_ = [('site_base', C.c_uint32), ('same_batch', C.c_uint8), ('reserved', C.c_uint8 * 3)]


class CommInfo(C.Structure):     # include/lgmi.h: lgmi_comm_info_t
    _fields_ = [('rccl_version', C.c_int32), ('nranks', C.c_int32), ('rank', C.c_int32), ('world_given', C.c_int32),
                ('rank_given', C.c_int32), ('initialised', C.c_int32), ('stand_in', C.c_int32), ('lib_path', C.c_char * 512)]


class SynthSpec(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('n_sites', C.c_uint32), ('n_reads', C.c_uint32),
                ('het_every', C.c_uint32), ('dropout_u16', C.c_uint32), ('het_noise_u16', C.c_uint32),
                ('tri_per_1024', C.c_uint32), ('tri_frac_u16', C.c_uint32), ('snp_per_1024', C.c_uint32),
                ('n_blocks', C.c_uint32), ('reserved', C.c_uint32)]


class LgmiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('liblgmi error %d: %s' % (code, msg))
        self.code = code


# every exported symbol of include/lgmi.h: name -> (restype, argtypes)
VP = C.c_void_p
SYMBOLS = {
    'lgmi_abi_version': (C.c_int, []),
    'lgmi_last_error': (C.c_char_p, []),
    'lgmi_struct_size': (C.c_size_t, [C.c_int]),
    'lgmi_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'lgmi_ctx_create': (C.c_int, [C.c_int, C.POINTER(VP)]),
    'lgmi_ctx_destroy': (None, [VP]),
    'lgmi_batch_upload': (C.c_int, [VP, C.POINTER(Batch), C.POINTER(VP)]),
    'lgmi_synth_dense': (C.c_int, [VP, C.POINTER(SynthSpec), C.POINTER(VP)]),
    'lgmi_dbatch_download': (C.c_int, [VP, C.POINTER(Batch)]),
    'lgmi_dbatch_free': (None, [VP]),
    'lgmi_run_device': (C.c_int, [VP, VP, C.POINTER(Params), C.POINTER(VP)]),
    'lgmi_run_device_rows': (C.c_int, [VP, VP, C.POINTER(Params), C.POINTER(VP)]),
    'lgmi_dresult_permute': (C.c_int, [VP, VP]),
    'lgmi_dresult_info': (C.c_int, [VP, C.POINTER(RunInfo)]),
    'lgmi_dresult_device_ptrs': (C.c_int, [VP, C.POINTER(Result)]),
    'lgmi_dresult_fetch': (C.c_int, [VP, C.POINTER(Result)]),
    'lgmi_dresult_fetch_compact': (C.c_int, [VP, C.POINTER(Result)]),
    'lgmi_result_expand_rows': (C.c_int, [C.POINTER(Result), u32p, u32p]),
    'lgmi_dresult_free': (None, [VP]),
    'lgmi_run': (C.c_int, [VP, C.POINTER(Batch), C.POINTER(Params), C.POINTER(Result), C.POINTER(RunInfo)]),
    'lgmi_result_free': (None, [C.POINTER(Result)]),
    'lgmi_site_mean': (C.c_int, [VP, C.c_uint64, u32p, u32p, f64p, C.c_uint64, f64p, u32p]),
    'lgmi_ecdf': (C.c_int, [VP, C.c_uint64, f64p, C.c_uint64, f64p, f64p]),
    'lgmi_plan_shard': (C.c_int, [C.POINTER(Batch), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(ShardPlan)]),
    'lgmi_shard_plan_free': (None, [C.POINTER(ShardPlan)]),
    'lgmi_ctx_synchronize': (C.c_int, [VP]),
    'lgmi_selftest_log': (C.c_int, [VP, C.c_uint64, f64p, f64p]),
    'lgmi_selftest_le_exp': (C.c_int, [VP, C.c_uint64, f64p, f64p, u8p, u8p, f64p, f64p]),
    'lgmi_comm_unique_id': (C.c_int, [VP]),
    'lgmi_comm_init': (C.c_int, [VP, VP, C.c_int, C.c_int]),
    'lgmi_comm_info': (C.c_int, [VP, C.POINTER(CommInfo)]),
    'lgmi_comm_allgather_u64': (C.c_int, [VP, C.c_uint64, u64p]),
    'lgmi_comm_allgather_u64v': (C.c_int, [VP, u64p, C.c_uint32, u64p]),
    'lgmi_comm_gather': (C.c_int, [VP, VP, C.c_int, C.POINTER(GatherOpts), C.POINTER(VP), u64p]),
    'lgmi_comm_gather_begin': (C.c_int, [VP, VP, C.c_int, C.POINTER(GatherOpts), C.POINTER(VP)]),
    'lgmi_comm_gather_finish': (C.c_int, [VP, C.POINTER(VP), u64p]),
    'lgmi_comm_gather_rows': (C.c_int, [VP, VP, C.c_int, C.POINTER(Result)]),
    'lgmi_comm_destroy': (None, [VP]),
}

_lib = None


def load():
    """dlopen liblgmi.so and bind every symbol; raises if the library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('liblgmi.so not found at %s — build it with `make -C l-giremi_amd` '
                               '(or __graft_entry__.build()); there is no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.lgmi_abi_version() != ABI_VERSION:
            raise RuntimeError('liblgmi.so ABI %d != binding ABI %d' % (lib.lgmi_abi_version(), ABI_VERSION))
        for which, st in enumerate((Batch, Params, Result, RunInfo, SynthSpec, ShardPlan, GatherOpts, CommInfo)):
            if lib.lgmi_struct_size(which) != C.sizeof(st):       # a stale declaration would be read / written past
                raise RuntimeError('liblgmi.so: sizeof %s is %d, the binding declares %d'
                                   % (st.__name__, lib.lgmi_struct_size(which), C.sizeof(st)))
        _lib = lib
    return _lib


def check(rc):
    if rc != OK:
        msg = load().lgmi_last_error().decode('utf-8', 'replace')
        if rc == E_DOMAIN:
            raise ValueError('math domain error')  # what the reference raises (log(0) in sklearn's MI)
        raise LgmiError(rc, msg)
