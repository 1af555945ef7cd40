"""Micro-benchmark of gt_conv_gemm_bf16 / gt_conv_wgrad_bf16 on the decoder shapes (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops, flow_impl
from glow_tts_amd.modules import ConvP
dev = torch.device("cuda:0")
B, T = 32, 400
rc = ops.RowsCtx(torch.full((B,), T, dtype=torch.int32, device=dev), T)
R = rc.R
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Cin, Cout, k, gate, name) in [(192, 384, 5, True, "in_layer+gate"), (192, 384, 1, False, "res_skip"), (192, 192, 1, False, "res only"),
                                   (80, 192, 1, False, "start"), (192, 160, 1, False, "end"), (384, 192, 5, False, "in_layer dgrad-like"),
                                   (192, 768, 3, False, "ffn1(R=dec)")]:
    x = torch.randn(R, Cin, device=dev).to(torch.bfloat16)
    conv = ConvP(Cin, Cout, k, gate=gate).to(dev); conv.prepare()
    if gate:
        f = lambda: ops.conv_rows(x, conv.pc, rc, bias=conv.bias, gate=True)
    else:
        f = lambda: ops.conv_rows(x, conv.pc, rc, bias=conv.bias)
    us = timeit(f)
    fl = 2.0 * R * Cin * Cout * k
    dy = torch.randn(R, Cout, device=dev).to(torch.bfloat16)
    g = lambda: flow_impl.conv_param_grads(conv, x, dy, R, want_bias=False)
    us2 = timeit(g)
    print(f"{name:22s} fwd {us:7.1f} us {fl/us/1e6:7.1f} TF | wgrad+wn_bwd {us2:7.1f} us {fl/us2/1e6:7.1f} TF", flush=True)
