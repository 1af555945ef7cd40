"""dev: intermediate tensors of the fused between-WaveNets kernels vs the five-kernel path (one or two blocks, eval mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from fill import fill_module
from glow_tts_amd import _lib, flow_impl, models, modules, ops

dev = torch.device("cuda:0")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, nb, 4, p_dropout=0.05), "decoder.").to(dev).eval()
modules.prepare_all(dec)
lens = [70, 33, 1, 64]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 70, lengths_host=lens, round_to=8)
g = torch.Generator().manual_seed(3)
rows = (torch.randn(rc.R, 160, generator=g)).to(dev) * rc.rowmask[:, None]
valid = rc.rowmask.bool()


def rel(a, b):
    a, b = a.float()[valid], b.float()[valid]
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


ld1 = torch.zeros(rc.B, device=dev)
z1, blocks = flow_impl.decoder_fwd_fused(rc, dec, rows, [None] * nb, ld1, False, 0)
ld2 = torch.zeros(rc.B, device=dev)
cur = rows
saved = []
for b in range(nb):
    an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
    y1, x0, s1 = flow_impl.actnorm_invconv_fwd(rc, cur, an.logs, an.bias, ic.weight, ld2)
    cur, s2 = flow_impl.coupling_fwd(rc, cb, y1, x0, None, ld2, False, 16 * b)
    saved.append((s1, s2))
    x, x0_, h0, wn_out, wn_saved, out = s2
    sv = blocks[b]
    print(f"block {b}: y {rel(sv.y, y1):.2e}  x0 {rel(sv.x0, x0):.2e}  h0 {rel(sv.h0, h0):.2e}  acts {rel(sv.wn_saved[3], wn_saved[3]):.2e}  "
          f"wn_out {rel(sv.wn_out, wn_out):.2e}  logs {rel(sv.logs_raw, out[:, 80:]):.2e}  z {rel(sv.z, cur):.2e}")
print("logdet", ld1.tolist(), ld2.tolist())

# backward
dz = (torch.randn(rc.R, 160, generator=g)).to(dev) * rc.rowmask[:, None]
dld = (torch.randn(rc.B, generator=g) * 0.1).to(dev)
from glow_tts_amd import wgrad
with wgrad.WgradQueue(dev, site=dec):
    dx1, g1, _ = flow_impl.decoder_bwd_fused(rc, dec, blocks, dz, dld, False)
g1 = {k: v.clone() for k, v in g1.items()}
g2 = {}
cur = dz
with wgrad.WgradQueue(dev, site=dec):
    for b in reversed(range(nb)):
        an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
        s1, s2 = saved[b]
        cur, gg, _ = flow_impl.coupling_bwd(rc, cb, s2, cur, dld, False)
        g2.update(gg)
        cur, gg = flow_impl.actnorm_invconv_bwd(rc, s1, cur, dld, an.logs, an.bias, ic.weight)
        g2.update(gg)
torch.cuda.synchronize()
print("dx", rel(dx1, cur))
names = {id(p): n for n, p in dec.named_parameters()}
for k in g2:
    a, b = g1[k].float(), g2[k].float()
    e = ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()
    if e > 5e-3:
        print("  grad", names[id(k)], f"{e:.2e}")
