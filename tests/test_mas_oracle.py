"""CPU tests: the oracle (oracle/mas_oracle.c) against the reference's own MAS.

Pins: (1) tests/golden/mas_golden.npz — produced by the compiled reference core.pyx
(tests/golden/make_mas_golden.py); (2) when oracle/_ref is present (build container, and the
GPU box via the prebuilt .so) a direct comparison on 320 random lattices.
"""
import hashlib
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from oracle import mas as omas  # noqa: E402
import make_mas_golden as mk  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mas_golden.npz")


def oracle_paths(values, t_x, t_y):
    v = np.ascontiguousarray(values.astype(np.float32)).copy()
    p = np.zeros(v.shape, dtype=np.int32)
    omas.oracle_maximum_path_c(p, v, t_x.astype(np.int32), t_y.astype(np.int32))
    return p


def test_oracle_matches_golden_small(built):
    g = np.load(GOLD)
    want = np.unpackbits(g["paths_packed"], axis=-1)[..., : g["values"].shape[-1]].astype(np.int32)
    got = oracle_paths(g["values"], g["t_x"], g["t_y"])
    assert got.shape == want.shape and len(got) >= 40
    assert np.array_equal(got, want)


def test_oracle_matches_golden_full_size(built):
    g = np.load(GOLD)
    for seed, digest in zip(g["full_seeds"], g["full_sha256"]):
        v, tx, ty = mk.full_size_lattice(int(seed))
        p = oracle_paths(v, tx, ty)
        assert hashlib.sha256(np.packbits(p.astype(np.uint8)).tobytes()).hexdigest() == str(digest)


def test_oracle_wrapper_semantics(built):
    """value*mask and lengths-from-mask (reference __init__.py:11,18-19)."""
    rng = np.random.default_rng(5)
    b, T_x, T_y = 3, 20, 41
    t_x = np.array([20, 7, 1]); t_y = np.array([41, 30, 1])
    mask = np.zeros((b, T_x, T_y), dtype=np.float32)
    for i in range(b):
        mask[i, : t_x[i], : t_y[i]] = 1
    value = rng.normal(size=(b, T_x, T_y)).astype(np.float32)
    p = omas.oracle_maximum_path(value, mask)
    assert p.dtype == np.int32
    for i in range(b):
        assert p[i].sum() == t_y[i]
        assert (p[i].sum(0)[: t_y[i]] == 1).all()
        assert (p[i].sum(1)[: t_x[i]] >= 1).all()
        assert p[i, t_x[i]:, :].sum() == 0 and p[i, :, t_y[i]:].sum() == 0
        assert p[i, 0, 0] == 1 and p[i, t_x[i] - 1, t_y[i] - 1] == 1


@pytest.mark.skipif(omas.ref_module() is None, reason="oracle/_ref (compiled reference core.pyx) absent")
def test_oracle_vs_compiled_reference_random(built):
    rng = np.random.default_rng(99)
    n = 0
    for trial in range(40):
        b = 8
        T_x = int(rng.integers(1, 70)); T_y = int(rng.integers(T_x, 140))
        t_x = rng.integers(1, T_x + 1, size=b).astype(np.int32)
        t_y = np.array([rng.integers(t_x[i], T_y + 1) for i in range(b)], dtype=np.int32)
        t_x[0], t_y[0] = T_x, T_y
        if trial % 3 == 0:
            v = rng.integers(-2, 2, size=(b, T_x, T_y)).astype(np.float32)     # tie-heavy
        elif trial % 3 == 1:
            v = rng.normal(-100, 5, size=(b, T_x, T_y)).astype(np.float32)
        else:
            v = rng.normal(-3e8, 2e8, size=(b, T_x, T_y)).astype(np.float32)   # crosses -1e9
        got = oracle_paths(v, t_x, t_y)
        vv = v.copy(); want = np.zeros(v.shape, dtype=np.int32)
        omas.ref_maximum_path_c(want, vv, t_x, t_y)
        assert np.array_equal(got, want), trial
        n += b
    assert n >= 300
