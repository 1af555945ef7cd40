"""Closed-form deterministic parameter values, regenerated identically on every side
(reference modules in make_float_golden.py, the oracle, the HIP-backed modules), so no weight
blobs are committed.  Ours — not reference code."""
import math
import zlib

import torch


def _phase(name):
    return (zlib.crc32(name.encode()) % 1000) / 1000.0 * 6.283185307179586


def closed_form(name, shape, dtype=torch.float32):
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    base = torch.sin(0.37 * i + _phase(name)) + 0.5 * torch.sin(0.011 * i * i % 6.283185307179586 + 1.3 * _phase(name))
    leaf = name.split(".")[-1]
    shape = tuple(shape)
    if leaf in ("weight_g",):
        v = 0.8 + 0.15 * base                       # positive gains
    elif leaf in ("gamma",):
        v = 1.0 + 0.1 * base
    elif leaf in ("logs",):
        v = 0.1 * base
    elif leaf in ("bias", "beta"):
        v = 0.1 * base
    elif leaf in ("emb_rel_k", "emb_rel_v"):
        v = 0.15 * base
    elif len(shape) == 2 and shape[0] == shape[1] and shape[0] <= 8:   # InvConvNear 4x4: well conditioned, det > 0
        v = (torch.eye(shape[0], dtype=torch.float64).reshape(-1) * 1.1 + 0.15 * base)
    elif len(shape) >= 2:
        fan_in = n // shape[0]
        v = base * (0.7 / math.sqrt(fan_in))
        if name.endswith("end.weight"):
            v = v * 0.3                             # keep coupling log-scales moderate
    else:
        v = 0.1 * base
    return v.reshape(shape).to(dtype)


def fill_module(module, prefix=""):
    """In-place closed-form fill of every parameter of an nn.Module (by its state-dict name)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if not p.requires_grad:                  # frozen tables (the fork's elevation / azimuth bin edges) keep their values
                continue
            p.copy_(closed_form(prefix + name, p.shape, p.dtype))
    return module


def filled_state(shapes, prefix=""):
    """dict name -> tensor for a {name: shape} table."""
    return {prefix + k: closed_form(prefix + k, s) for k, s in shapes.items()}
