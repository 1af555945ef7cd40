"""Micro-benchmark of gt_attn_fwd / gt_attn_bwd (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.lib()
B, T, H, D = 32, int(os.environ.get("T", 150)), 2, 96
rc = ops.RowsCtx(torch.randint(T // 2, T + 1, (B,), dtype=torch.int32).to(dev), T)
R, C = rc.R, H * D
q, k, v, do = [torch.randn(R, C, device=dev).to(torch.bfloat16) for _ in range(4)]
Ek = torch.randn(9, D, device=dev) * 0.1; Ev = torch.randn(9, D, device=dev) * 0.1
o = torch.zeros(R, C, dtype=torch.bfloat16, device=dev); P = torch.empty(B, H, T, T, device=dev)
wsb = L.gt_attn_bwd_workspace_bytes(B, T, H); dS = torch.empty(wsb, dtype=torch.uint8, device=dev); dq, dk, dv = [torch.zeros(R, C, dtype=torch.bfloat16, device=dev) for _ in range(3)]
dEk = torch.zeros_like(Ek); dEv = torch.zeros_like(Ev)
st = _lib.current_stream(dev)
def fwd():
    assert L.gt_attn_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths), _lib.ptr(o), C, _lib.ptr(P),
                         B, T, rc.Tp, None, H, D, 4, 0.1, 7, None, st) == 0
def bwd():
    assert L.gt_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths), _lib.ptr(do), C, _lib.ptr(P), _lib.ptr(dS), wsb,
                         _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), C, _lib.ptr(dEk), _lib.ptr(dEv), B, T, rc.Tp, None, H, D, 4, 0.1, 7, None, st) == 0
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print(f"T={T} fwd {timeit(fwd):.1f} us  bwd {timeit(bwd):.1f} us", flush=True)
