"""Micro-benchmark of gt_layernorm_bwd through the encoder helper, graph-replayed (dev tool; GT_LNB = waves*1000 + rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from glow_tts_amd import ops, encoder_impl
from glow_tts_amd.modules import LayerNorm
dev = torch.device("cuda:0")
R, C = 3584, 192
lens = torch.full((28,), 124, dtype=torch.int32, device=dev)
rc = ops.RowsCtx(lens, 124)
R = rc.R
ln = LayerNorm(C).to(dev)
a = torch.randn(R, C, device=dev); y = torch.randn(R, C, device=dev).to(torch.bfloat16)
x, xb, saved = encoder_impl._ln_fwd(rc, ln, a, y, 0.1, 1, 0.0, 0, 0, True, C)
dout = torch.randn(R, C, device=dev); doutb = torch.randn(R, C, device=dev).to(torch.bfloat16)
def f():
    g = {}
    encoder_impl._ln_bwd(rc, ln, saved, dout, doutb, True, True, g)
f(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    f()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20): f()
torch.cuda.synchronize()
for _ in range(3): gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): gr.replay()
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'} R={R} C={C}: {e0.elapsed_time(e1)/400*1e3:.1f} us per call (incl. 2 arena/zero allocs)")
