"""dev: where a WaveNet stack launch spends its time — shader-clock stamps of wave 0 of every workgroup (a -DWNS_PHASES=1 build of
csrc/wn_stack.hip made by tools/exp_variant.py), median over workgroups, in cycles and as a share of the launch.

    python tools/exp_variant.py ph wn_stack -DWNS_PHASES=1
    python tools/wn_stack_phases.py glow-tts_amd/build/exp/libglowtts_ph.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glow_tts_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from glow_tts_amd import modules, ops, flow_impl, wgrad

dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
H, n = 192, 4
wn = modules.WN(160, H, 5, 1, n, 0, 0.05).to(dev)
modules.prepare_all(wn)
g = torch.Generator().manual_seed(1234)
t_y = (torch.randint(150, 401, (32,), generator=g) * 2); t_y[0] = 800
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 400, lengths_host=lens, round_to=512)
R = rc.R
x = (torch.randn(R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
dsk = (torch.randn(R, H, device=dev) * rc.rowmask[:, None] * 0.1).to(torch.bfloat16)
via_all = (torch.randn(R, n * H, device=dev) * rc.rowmask[:, None] * 0.1).to(torch.bfloat16)
nwg = (R + 51) // 52


def phases(label, names):
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 48, dtype=np.uint64)
    assert raw.gt_dev_wns_phases(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    ph = buf.reshape(1024, 48)[:nwg].astype(np.int64)
    idx = [i for i, _ in names]
    t = ph[:, idx] - ph[:, [0]]
    med = np.median(t, axis=0)
    total = med[-1]
    print(f"-- {label}: {nwg} workgroups, launch = {total:.0f} cycles of wave 0 (median); start skew p5..p95 = "
          f"{np.percentile(ph[:, 0] - ph[:, 0].min(), [5, 95])}")
    prev = 0.0
    for (i, nm), m in zip(names, med):
        print(f"   {nm:46s} {m - prev:8.0f}  ({(m - prev) / total * 100:5.1f} %)   at {m:8.0f}")
        prev = m


for _ in range(3):
    saved = flow_impl.wn_fwd(rc, wn, x, None, True, 7, layers_only=True)[1]
fn = [(1, "x0 tile load + barrier")]
for l in range(n):
    fn += [(2 + 6 * l, f"L{l} k=5 conv loop (360 MFMA / wave)"), (3 + 6 * l, f"L{l} gate epilogue -> LDS tiles"), (4 + 6 * l, f"L{l} barrier")]
    if l < n - 1:
        fn += [(5 + 6 * l, f"L{l} residual 1x1 (36 MFMA / wave)"), (6 + 6 * l, f"L{l} next ring, T/S/acts stores, x_next epilogue, barrier"),
               (7 + 6 * l, f"L{l} x_next stores issued")]
fn += [(40, "last layer's T/S/acts stores issued, end")]
phases("forward", fn)

for _ in range(3):
    q = wgrad.WgradQueue(dev, site=wn); q.__enter__()
    flow_impl.wn_bwd(rc, wn, saved, dsk, dacts_skip=via_all)
    wgrad._ACTIVE.pop(); q.items = []
bn = [(1, "head: top-layer gate backward on 68 rows + barrier")]
for J in (3, 2, 1, 0):
    b = 8 * (3 - J)
    bn += [(2 + b, f"J{J} conv^T loop (180 MFMA / wave)"), (3 + b, f"J{J} barrier, K-half exchange through LDS, sum"), (4 + b, f"J{J} dX epilogue -> tile, barrier")]
    if J > 0:
        bn += [(5 + b, f"J{J} residual 1x1^T (36 MFMA / wave)"), (6 + b, f"J{J} next ring, dX stores, operands -> LDS, barrier"),
               (7 + b, f"J{J} gate backward epilogue"), (8 + b, f"J{J} barrier"), (9 + b, f"J{J} d pre stores issued")]
bn += [(40, "dX_0 stores issued, end")]
phases("backward", bn)
