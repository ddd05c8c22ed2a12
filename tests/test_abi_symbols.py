"""CPU-only checks of the drop-in boundary: liblgmi.so loads without a GPU, exports
every symbol include/lgmi.h declares, and refuses to work without a device (no
CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from lgmi import _lib


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'lgmi.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(lgmi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), 'liblgmi.so does not export %s' % name
    assert sorted(_lib.SYMBOLS) == names, 'ctypes binding and header disagree'


def test_io_library_exports_every_declared_symbol():
    """liblgmi_io.so (host-only BAM reader) against include/lgmi_io.h and the ctypes table in lgmi/io.py"""
    from lgmi import io
    text = open(os.path.join(ROOT, 'include', 'lgmi_io.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = sorted(set(re.findall(r'\b(lgio_[a-z0-9_]+)\s*\(', text)))
    lib = C.CDLL(io.IO_LIB_PATH)
    assert len(names) >= 14
    for name in names:
        assert hasattr(lib, name), 'liblgmi_io.so does not export %s' % name
    assert sorted(io.IO_SYMBOLS) == names
    assert io.load_io().lgio_abi_version() == 4
    assert C.sizeof(io._Reads) == 152 and C.sizeof(io._Pileup) == 48 + 152


def test_abi_version_and_struct_sizes():
    lib = _lib.load()
    assert lib.lgmi_abi_version() == _lib.ABI_VERSION
    assert C.sizeof(_lib.Batch) == 96
    assert C.sizeof(_lib.Params) == 32
    assert C.sizeof(_lib.Result) == 144
    assert C.sizeof(_lib.RunInfo) == 128
    assert C.sizeof(_lib.ShardPlan) == 152
    assert C.sizeof(_lib.GatherOpts) == 8
    assert C.sizeof(_lib.SynthSpec) == 48


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.lgmi_ctx_create(0, C.byref(h))
    assert rc == _lib.E_NODEV and not h
    assert b'no CPU fallback' in lib.lgmi_last_error()
    with pytest.raises(_lib.LgmiError):
        import lgmi
        lgmi.Engine()


def test_product_never_imports_the_oracle():
    """the oracle is the checker: nothing in the product package may import, link or execute it
    (comments may cite the specification file by name)"""
    pkg = os.path.join(ROOT, 'l-giremi_amd')
    for dirpath, _dirs, files in os.walk(pkg):
        for fn in files:
            if not fn.endswith(('.py', '.cpp', '.hip', '.h', 'Makefile')):
                continue
            src = open(os.path.join(dirpath, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), '%s imports the oracle' % fn
            for token in ('liblgmi_oracle', 'c_oracle', 'mi_oracle', 'lgo_'):
                assert token not in src, '%s references %s' % (fn, token)
