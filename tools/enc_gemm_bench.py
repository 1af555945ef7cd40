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

if len(sys.argv) > 1 and sys.argv[1] == "cold":
    # the same launches behind a 640 MB fill (L2 and the Infinity Cache evicted: the state of the text encoder's backward inside the step),
    # with and without the weight image touched again behind the fill: what the first touch of the weights costs a short GEMM launch
    big = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device=dev)
    fill = timeit(lambda: big.fill_(1.0), n=10, reps=5)
    print(f"640 MB fill alone {fill:6.1f} us")
    for (Cin, Cout, k, name) in [(192, 576, 1, "qkv fused"), (192, 768, 3, "ffn conv1"), (768, 192, 3, "ffn conv2")]:
        x = torch.randn(R, Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(R, Cout, device=dev).to(torch.bfloat16)
        pc = ops.PackedConv(Cout, Cin, k).pack(torch.randn(Cout, Cin, k, device=dev) * 0.05)
        warm = timeit(lambda: ops.conv_rows(dy, pc, rc, dgrad=True))
        cold = timeit(lambda: (big.fill_(1.0), ops.conv_rows(dy, pc, rc, dgrad=True)), n=10, reps=5) - fill
        touch = timeit(lambda: (big.fill_(1.0), pc.dgrad.view(torch.int32).sum()), n=10, reps=5) - fill
        both = timeit(lambda: (big.fill_(1.0), pc.dgrad.view(torch.int32).sum(), ops.conv_rows(dy, pc, rc, dgrad=True)), n=10, reps=5) - fill - touch
        print(f"{name:14s} dgrad: warm {warm:5.1f} us, cold {cold:5.1f} us, cold with the weights touched first {both:5.1f} us", flush=True)
