"""Micro-benchmark of gt_actnorm_invconv_bwd, graph-replayed (dev tool; GT_ANB_ROWS = rows per workgroup)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops, flow_impl
dev = torch.device("cuda:0")
lens = torch.full((32,), 284, dtype=torch.int32, device=dev)
rc = ops.RowsCtx(lens, 284)
R, C = rc.R, 160
x = torch.randn(R, C, device=dev); dy = torch.randn(R, C, device=dev)
logs = torch.randn(1, C, 1, device=dev) * 0.1; bias = torch.randn(1, C, 1, device=dev) * 0.1
W = torch.linalg.qr(torch.randn(4, 4))[0].to(dev)
logdet = torch.zeros(32, device=dev); dlogdet = torch.ones(32, device=dev)
y, x0, saved = flow_impl.actnorm_invconv_fwd(rc, x, logs, bias, W, logdet)
def f():
    flow_impl.actnorm_invconv_bwd(rc, saved, dy, dlogdet, logs, bias, W)
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
print(f"GT_ANB_ROWS={os.environ.get('GT_ANB_ROWS','default')} R={R}: {e0.elapsed_time(e1)/400*1e3:.1f} us per call (incl. 3 zero fills + logdet_bwd)")
