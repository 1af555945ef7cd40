"""dev: latency of gt_conv_gemm_bf16 on the text encoder's shapes (R ~ 4k rows), per tile variant: enc_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops
dev = torch.device("cuda:0")
B, T = 32, 120
rc = ops.RowsCtx(torch.full((B,), T, dtype=torch.int32, device=dev), T)
R = rc.R
def timeit(fn, n=20, reps=10):
    """n launches captured in one HIP graph (eager launches from Python cost ~11 us each, more than these kernels)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
print("rows", R)
for (Cin, Cout, k, name) in [(192, 192, 1, "attn q/k/v/o"), (192, 576, 1, "qkv fused"), (192, 768, 3, "ffn conv1"), (768, 192, 3, "ffn conv2"),
                             (192, 192, 5, "prenet"), (192, 256, 3, "dp conv1")]:
    x = torch.randn(R, Cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(R, Cout, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, k, device=dev) * 0.05
    pc = ops.PackedConv(Cout, Cin, k).pack(w)
    bias = torch.zeros(Cout, device=dev)
    out = []
    for tile, tn in ((0, "auto"), (1, "64x64"), (2, "64x128"), (6, "taps")):
        if tile == 2 and (pc.Np_f % 128 or pc.Np_d % 128):
            continue
        f = timeit(lambda: ops.conv_rows(x, pc, rc, bias=bias, tile=tile))
        d = timeit(lambda: ops.conv_rows(dy, pc, rc, dgrad=True, tile=tile))
        out.append(f"{tn} fwd {f:5.1f} dgrad {d:5.1f}")
    print(f"{name:14s} " + " | ".join(out), flush=True)
