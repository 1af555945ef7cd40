"""Where does the conv GEMM spend its time?  Launches are replayed from a HIP graph so the host launch cost
(~10 us per call from Python) does not hide the kernel (dev tool; GT_CONV_EXP bits: 1 = no epilogue, 2 = no in-loop loads)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import ops
from glow_tts_amd.modules import ConvP
dev = torch.device("cuda:0")
NL = 20
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(NL): fn()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / NL * 1e3
shapes = [(192, 384, 5, True, 0.05, "gate p=.05"), (192, 384, 5, True, 0.0, "gate p=0"), (192, 384, 5, False, 0.0, "plain N=384"),
          (384, 192, 5, False, 0.0, "dgrad-like"), (192, 192, 1, False, 0.0, "1x1 K=192"), (384, 192, 1, False, 0.0, "1x1 K=384"),
          (192, 384, 1, False, 0.0, "1x1 N=384")]
for B, T in [(int(os.environ.get("B", 23)), 400)]:
    rc = ops.RowsCtx(torch.full((B,), T, dtype=torch.int32, device=dev), T)
    R = rc.R
    for (Cin, Cout, k, gate, p, name) in shapes:
        x = torch.randn(R, Cin, device=dev).to(torch.bfloat16)
        conv = ConvP(Cin, Cout, k, gate=gate).to(dev); conv.prepare()
        y = torch.empty(R, Cout // 2 if gate else Cout, dtype=torch.bfloat16, device=dev)
        if gate:
            t = torch.empty_like(y); s_ = torch.empty_like(y)
            f = lambda: ops.conv_rows(x, conv.pc, rc, bias=conv.bias, gate=True, drop_p=p, seed=1, out=y, gate_t=t, gate_s=s_)
        else:
            f = lambda: ops.conv_rows(x, conv.pc, rc, bias=conv.bias, out=y)
        us = timeit(f)
        nwg = -(-R // 128) * (-(-Cout // (128 if Cout % 128 == 0 else 64)))
        print(f"R={R:6d} wgs={nwg:4d} {name:12s} {us:7.1f} us {2.0*R*Cin*Cout*k/us/1e6:7.1f} TF", flush=True)
