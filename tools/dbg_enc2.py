import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests/golden"); sys.path.insert(0, ROOT + "/tests")
import torch, torch.nn.functional as F
from fill import fill_module
from oracle import glowtts_ref as R
from test_encoder_gpu import cpu_state, lens_mask, relerr
from glow_tts_amd import attentions, encoder_impl
dev = torch.device("cuda:0")
enc = fill_module(attentions.Encoder(192, 768, 2, 1, 3, 0.1, window_size=4), "enc.").eval()
P = cpu_state(enc, "enc.")
T, lens = 41, [41, 17, 30]
xm = lens_mask(lens, T)
g = torch.Generator().manual_seed(2)
x = torch.randn(3, 192, T, generator=g) * xm
xx = x.clone().requires_grad_(True)
# oracle layer 0 inline, keeping intermediates
am = xm.unsqueeze(2) * xm.unsqueeze(-1)
y, _ = R.mha_fwd(P, "enc.attn_layers.0.", xx * xm, xx * xm, am)
x1 = R.layer_norm_c(xx * xm + y, P["enc.norm_layers_1.0.gamma"], P["enc.norm_layers_1.0.beta"])
c1 = R.conv1d(P, "enc.ffn_layers.0.conv_1", x1 * xm, padding=1); c1.retain_grad()
h = torch.relu(c1)
f2 = R.conv1d(P, "enc.ffn_layers.0.conv_2", h * xm, padding=1) * xm; f2.retain_grad()
x2 = R.layer_norm_c(x1 + f2, P["enc.norm_layers_2.0.gamma"], P["enc.norm_layers_2.0.beta"])
o = x2 * xm
r = torch.randn(o.shape, generator=g)
(o * r).sum().backward()
rec = {}
orig = encoder_impl.conv_param_grads
def spy(conv, xr, dy, Rr, want_bias=True):
    rec[id(conv)] = (xr.clone(), dy.clone())
    return orig(conv, xr, dy, Rr, want_bias)
encoder_impl.conv_param_grads = spy
enc = enc.to(dev)
xd = x.to(dev).requires_grad_(True)
od = enc(xd, xm.to(dev))
(od * r.to(dev)).sum().backward()
from glow_tts_amd import ops
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), T)
xr, dc1 = rec[id(enc.ffn_layers[0].conv_1)]
_, df2 = rec[id(enc.ffn_layers[0].conv_2)]
print("out", relerr(od.detach().cpu(), o.detach()))
print("df2 vs oracle", relerr(rc.from_rows(df2).float().cpu(), f2.grad))
got = rc.from_rows(dc1).float().cpu(); want = c1.grad * xm
print("dc1 vs oracle", relerr(got, want))
d = (got - want).abs()
print("dc1 err by frame (utt0):", [round(v, 2) for v in d[0].amax(dim=0).tolist()])
print("dc1 err by frame (utt1):", [round(v, 2) for v in d[1].amax(dim=0).tolist()])
print("x1 in vs oracle", relerr(rc.from_rows(xr).float().cpu(), (x1 * xm).detach()))
