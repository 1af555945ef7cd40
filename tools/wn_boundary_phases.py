"""dev: where a between-WaveNets launch spends its time — shader-clock stamps of wave 0 of every workgroup (a -DWNB_PHASES=1 build of
csrc/wn_boundary.hip made by tools/exp_variant.py), median over workgroups.

    python tools/exp_variant.py phb wn_boundary -DWNB_PHASES=1
    python tools/wn_boundary_phases.py glow-tts_amd/build/exp/libglowtts_phb.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glow_tts_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from glow_tts_amd import flow_impl, models, modules, ops, wgrad

dev = torch.device("cuda:0")
L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
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
nwg = (rc.R + 63) // 64
fw = [t for t in trace if t[0].endswith("fwd")]
bw = [t for t in trace if t[0].endswith("bwd")]


def phases(label, call, names):
    fn, args, _ = call
    for _ in range(3):
        _lib.check(getattr(L, fn)(ctypes.byref(args), _lib.current_stream(dev)), fn)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 48, dtype=np.uint64)
    assert raw.gt_dev_wnb_phases(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    ph = buf.reshape(1024, 48)[:nwg].astype(np.int64)
    t = ph[:, [i for i, _ in names]] - ph[:, [0]]
    med = np.median(t, axis=0)
    print(f"-- {label}: {nwg} workgroups, {med[-1]:.0f} cycles of wave 0 from its first stamp (median)")
    prev = 0.0
    for (i, nm), m in zip(names, med):
        print(f"   {nm:58s} {m - prev:8.0f}  ({(m - prev) / med[-1] * 100:5.1f} %)   at {m:8.0f}")
        prev = m


phases("forward, tail of block b + head of block b + 1", fw[5],
       [(1, "acts loads + first skip fragments issued"), (2, "skip GEMM, K = 768 (4 slices: regs -> LDS, barrier, 36 MFMA)"),
        (3, "skip epilogue -> tile, barrier"), (4, "wn_out store, y loads issued, end conv (36 MFMA)"), (5, "end epilogue -> fp32 tile, barrier"),
        (6, "coupling (+ z, logs stores), barrier"), (7, "log-det atomics"), (8, "ActNorm + InvConvNear (+ y, y0 stores), barrier"),
        (9, "start conv (15 MFMA)"), (10, "start epilogue -> tile, barrier"), (11, "h store, end")])
phases("backward, head of block b + 1 + tail of block b", bw[5],
       [(1, "d h tile load + barrier"), (2, "start data gradient (24 MFMA) + epilogue + barrier"), (3, "ActNorm / InvConvNear backward (16 rows per wave)"),
        (4, "parameter-gradient fold + atomics"), (5, "coupling backward (+ dx, dout stores), barrier"), (6, "end data gradient (30 MFMA)"),
        (7, "epilogue -> tile, barrier"), (8, "d wn_out store"), (9, "skip window 0 (36 MFMA) + store"), (10, "skip window 1"), (11, "skip window 2"),
        (12, "skip window 3, end")])
