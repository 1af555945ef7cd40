import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from fill import fill_module
from glow_tts_amd import _lib, flow_impl, models, modules, ops
dev = torch.device("cuda:0")
dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 1, 4, p_dropout=0.05), "decoder.").to(dev).eval()
modules.prepare_all(dec)
lens = [70, 33, 1, 64]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 70, lengths_host=lens, round_to=8)
g = torch.Generator().manual_seed(3)
rows = (torch.randn(rc.R, 160, generator=g)).to(dev) * rc.rowmask[:, None]
ld1 = torch.zeros(rc.B, device=dev)
z1, blocks = flow_impl.decoder_fwd_fused(rc, dec, rows, [None], ld1, False, 0)
an, ic = dec.flows[0], dec.flows[1]
ld2 = torch.zeros(rc.B, device=dev)
y1, x0, s1 = flow_impl.actnorm_invconv_fwd(rc, rows, an.logs, an.bias, ic.weight, ld2)
d = (blocks[0].y - y1).abs()
print("R", rc.R, "max diff per row (first 80):", [round(v, 3) for v in d.max(1).values[:80].tolist()])
print("max diff per col:", [round(v, 3) for v in d.max(0).values.tolist()])
x = rows[2]
lg, bs, W = an.logs.reshape(-1), an.bias.reshape(-1), ic.weight
a = bs + torch.exp(lg) * x
ref = torch.zeros(160, device=dev)
for gq in range(40):
    ch = [2 * gq, 2 * gq + 1, 80 + 2 * gq, 80 + 2 * gq + 1]
    av = a[ch]
    o = W @ av
    for k in range(4):
        ref[ch[k]] = o[k]
print("row2 torch ref   ", ref[:6].tolist(), ref[80:84].tolist())
print("row2 fused       ", blocks[0].y[2, :6].tolist(), blocks[0].y[2, 80:84].tolist())
print("row2 five-kernel ", y1[2, :6].tolist(), y1[2, 80:84].tolist())
print("x row2", x[:6].tolist())
print("logs", lg[:4].tolist(), "bias", bs[:4].tolist(), "W", W.reshape(-1).tolist())
print("initialized", an.initialized)
