"""dev: the decoder's deferred weight-gradient flush alone, at the bench shapes (cfg 2: 12 blocks x 4 WaveNet layers, ~8.9 k rows), every
job on its own x / dy rows (0.5 GB of saved rows: cold, as in the step).  Prints the time of the k = 5 launch, the k = 1 launch and the
weight-norm backward, graph-replayed:  wgrad_bench.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from glow_tts_amd import flow_impl, wgrad
from glow_tts_amd.modules import ConvP, WNConvP
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8896
torch.manual_seed(0)

def rows(C):
    return (torch.randn(R, C, device=dev) * 0.5).to(torch.bfloat16)

def make(kind):
    convs, ops_ = [], []
    for blk in range(12):
        if kind in ("k5", "all"):
            for l in range(4):
                c = WNConvP(192, 384, 5).to(dev); c.prepare(); convs.append((c, rows(192), rows(384)))
        if kind in ("k1", "all"):
            for l in range(3):
                c = WNConvP(192, 384, 1).to(dev); c.prepare(); convs.append((c, rows(192), rows(384)))
            c = WNConvP(192, 192, 1).to(dev); c.prepare(); convs.append((c, rows(192), rows(192)))
            c = WNConvP(80, 192, 1).to(dev); c.prepare(); convs.append((c, rows(80), rows(192)))
            c = ConvP(192, 160, 1).to(dev); c.prepare(); convs.append((c, rows(192), rows(160)))
    return convs

def run(convs):
    with wgrad.WgradQueue(dev):
        for c, x, dy in convs:
            flow_impl.conv_param_grads(c, x, dy, R)

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for kind in ("k5", "k1", "all"):
    convs = make(kind)
    us = timeit(lambda: run(convs))
    print(f"{kind:4s} {len(convs):3d} jobs, {R} rows: {us:8.1f} us per flush (wgrad launches + weight-norm backward; eager, host planning cached)", flush=True)
    del convs
