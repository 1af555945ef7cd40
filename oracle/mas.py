"""TEST INFRASTRUCTURE ONLY — Python handles on the MAS oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

* ``oracle_maximum_path_c``      — our C restatement (oracle/mas_oracle.c), follows
                                   reference monotonic_align/core.pyx:9-45.
* ``ref_maximum_path_c``         — the reference's own core.pyx, compiled by
                                   oracle/Makefile into oracle/_ref/ (None if absent).
* ``oracle_maximum_path``        — numpy restatement of the copy-heavy wrapper
                                   reference monotonic_align/__init__.py:6-21.
"""
import ctypes
import glob
import importlib.util
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_mas.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle_mas.so missing — run `make -C oracle` "
                               "(or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f32p = ctypes.POINTER(ctypes.c_float)
        lib.mas_oracle_batch.argtypes = [i32p, f32p, i32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.mas_oracle_batch.restype = None
        lib.mas_oracle_range.argtypes = [i32p, f32p, i32p, i32p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int]
        lib.mas_oracle_range.restype = None
        lib.mas_oracle_batch_omp.argtypes = [i32p, f32p, i32p, i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.mas_oracle_batch_omp.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def oracle_maximum_path_c(paths, values, t_xs, t_ys):
    """In-place, like reference core.pyx:38-45: `values` is mutated, `paths` (pre-zeroed
    int32) receives the 0/1 path.  All arrays C-contiguous."""
    assert paths.dtype == np.int32 and values.dtype == np.float32
    assert t_xs.dtype == np.int32 and t_ys.dtype == np.int32
    assert paths.flags.c_contiguous and values.flags.c_contiguous
    b, T_x, T_y = values.shape
    _lib().mas_oracle_batch(_ptr(paths, ctypes.c_int32), _ptr(values, ctypes.c_float),
                            _ptr(t_xs, ctypes.c_int32), _ptr(t_ys, ctypes.c_int32), b, T_x, T_y)


def oracle_maximum_path_omp(paths, values, t_xs, t_ys, n_threads=0):
    """oracle_maximum_path_c with one utterance per OpenMP thread (n_threads = 0: all cores); returns the threads used."""
    assert paths.dtype == np.int32 and values.dtype == np.float32 and t_xs.dtype == np.int32 and t_ys.dtype == np.int32
    assert paths.flags.c_contiguous and values.flags.c_contiguous
    b, T_x, T_y = values.shape
    return _lib().mas_oracle_batch_omp(_ptr(paths, ctypes.c_int32), _ptr(values, ctypes.c_float), _ptr(t_xs, ctypes.c_int32),
                                       _ptr(t_ys, ctypes.c_int32), b, T_x, T_y, int(n_threads))


def oracle_maximum_path_range(paths, values, t_xs, t_ys, i0, i1):
    b, T_x, T_y = values.shape
    _lib().mas_oracle_range(_ptr(paths, ctypes.c_int32), _ptr(values, ctypes.c_float),
                            _ptr(t_xs, ctypes.c_int32), _ptr(t_ys, ctypes.c_int32),
                            int(i0), int(i1), T_x, T_y)


def ref_module():
    """The compiled reference core.pyx (oracle/_ref/core*.so) or None."""
    global _REF
    if _REF is None:
        hits = glob.glob(os.path.join(_HERE, "_ref", "core*.so"))
        if not hits:
            _REF = False
        else:
            spec = importlib.util.spec_from_file_location("core", hits[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _REF = mod
    return _REF or None


def ref_maximum_path_c(paths, values, t_xs, t_ys):
    mod = ref_module()
    if mod is None:
        raise RuntimeError("oracle/_ref reference build absent")
    mod.maximum_path_c(paths, values, t_xs, t_ys)


def oracle_maximum_path(value, mask, core=None):
    """numpy restatement of reference monotonic_align/__init__.py:6-21 (CPU arrays in,
    int32 path out).  `core` selects the C routine (default: our restatement)."""
    core = core or oracle_maximum_path_c
    value = (value * mask).astype(np.float32)            # :11, :14
    path = np.zeros_like(value).astype(np.int32)         # :15
    t_x_max = mask.sum(1)[:, 0].astype(np.int32)         # :18
    t_y_max = mask.sum(2)[:, 0].astype(np.int32)         # :19
    value = np.ascontiguousarray(value)
    core(path, value, np.ascontiguousarray(t_x_max), np.ascontiguousarray(t_y_max))
    return path
