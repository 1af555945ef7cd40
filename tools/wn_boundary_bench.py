"""dev: per-launch time of the fused between-WaveNets kernels (csrc/wn_boundary.hip), graph-replayed on cfg2-shaped rows:
the launches of one real decoder forward + backward are recorded (flow_impl.BOUNDARY_TRACE) and replayed back to back."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib, flow_impl, models, modules, ops, wgrad

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
L = _lib.lib()
nb = 12
dec = models.FlowSpecDecoder(80, 192, 5, 1, nb, 4, p_dropout=0.05).to(dev).train()
for b in range(nb):
    torch.nn.init.normal_(dec.flows[3 * b + 2].end.weight, std=0.01)
modules.prepare_all(dec)
g = torch.Generator().manual_seed(1234)
t_y = (torch.randint(150, 401, (32,), generator=g) * 2); t_y[0] = 800
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 400, lengths_host=lens, round_to=512)
rows = torch.randn(rc.R, 160, device=dev) * rc.rowmask[:, None]
flow_impl.BOUNDARY_TRACE = trace = []
ld = torch.zeros(rc.B, device=dev)
z, blocks = flow_impl.decoder_fwd_fused(rc, dec, rows, [None] * nb, ld, True, 7)
dld = torch.zeros(rc.B, device=dev)
with wgrad.WgradQueue(dev, site=dec):
    flow_impl.decoder_bwd_fused(rc, dec, blocks, torch.randn_like(rows) * rc.rowmask[:, None], dld, False)
torch.cuda.synchronize()
flow_impl.BOUNDARY_TRACE = None
st = lambda: _lib.current_stream(dev)


def timeit(name, calls, replays=20):
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for fn, args, _ in calls[:1]:
            _lib.check(getattr(L, fn)(ctypes.byref(args), st()), fn)
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=stream):
            for fn, args, _ in calls:
                _lib.check(getattr(L, fn)(ctypes.byref(args), st()), fn)
    torch.cuda.synchronize()
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / (replays * len(calls)) * 1e3:7.1f} us / launch  ({len(calls)} launches, R = {rc.R})", flush=True)


fw = [t for t in trace if t[0].endswith("fwd")]
bw = [t for t in trace if t[0].endswith("bwd")]
timeit("fwd tail + head (blocks 1..11)", fw[1:-1])


def variant(calls, cls, **over):
    out = []
    for fn, args, kw in calls:
        k2 = dict(kw); k2.update(over(kw) if callable(over) else over)
        out.append((fn, _lib.fill_args(cls, **{k: v for k, v in k2.items() if v is not None}), k2))
    return out


timeit("fwd head only (x_in = previous z)", variant(fw[1:-1], _lib.BoundaryFwdArgs, over=None) if False else
       [(fn, _lib.fill_args(_lib.BoundaryFwdArgs, **{k: v for k, v in dict(kw, acts=None, x_in=kw["z"]).items() if v is not None}), kw) for fn, a, kw in fw[1:-1]])
timeit("fwd tail only", [(fn, _lib.fill_args(_lib.BoundaryFwdArgs, **{k: v for k, v in dict(kw, y_next=None).items() if v is not None}), kw) for fn, a, kw in fw[1:-1]])
timeit("bwd head + tail", bw[1:-1])
timeit("bwd tail only (dz_in = dx_in)", [(fn, _lib.fill_args(_lib.BoundaryBwdArgs, **{k: v for k, v in dict(kw, dh=None, dz_in=kw["dx_in"]).items() if v is not None}), kw) for fn, a, kw in bw[1:-1]])
timeit("bwd head only", [(fn, _lib.fill_args(_lib.BoundaryBwdArgs, **{k: v for k, v in dict(kw, dout=None).items() if v is not None}), kw) for fn, a, kw in bw[1:-1]])
