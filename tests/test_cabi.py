"""CPU tests: the C-ABI shared library loads and exports every symbol include/*.h declares
(no compute calls without a GPU), and the Python binding table matches the header."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = open(h).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(gt_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_header_declares_something():
    syms = declared_symbols()
    assert "gt_mas_f32" in syms and "gt_version" in syms


def test_library_exports_every_declared_symbol(built):
    from glow_tts_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} declared in include/ but not exported"


def test_binding_table_matches_header(built):
    from glow_tts_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    L = _lib.lib()
    assert b"gfx950" in L.gt_version()


def test_host_helpers_no_gpu(built):
    from glow_tts_amd import _lib
    L = _lib.lib()
    # LDS sizing helper is pure host code
    assert L.gt_mas_lds_bytes(150, 800) < 160 * 1024
    assert L.gt_mas_lds_bytes(375, 870) < 160 * 1024
    assert L.gt_mas_lds_bytes(0, 10) == 0
    assert L.gt_mas_workspace_bytes(32, 150, 800) >= 32 * 151 * 4
    # argument validation happens before any launch
    assert L.gt_mas_f32(None, None, None, None, None, 0, None, None, 2, 8, 8, 64, 8, None, 0, None, None) == -1
    assert L.gt_mas_f32(None, None, None, None, None, 0, None, None, 0, 8, 8, 64, 8, None, 0, None, None) == 0
    assert L.gt_mas_f32(None, None, None, None, None, 0, None, None, -1, 8, 8, 64, 8, None, 0, None, None) == -1


def test_ops_fail_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from glow_tts_amd import monotonic_align
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        monotonic_align.maximum_path(torch.zeros(1, 2, 3), torch.ones(1, 2, 3))
