import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests/golden"); sys.path.insert(0, ROOT + "/tests")
import torch
from fill import fill_module
from oracle import glowtts_ref as R
from test_encoder_gpu import cpu_state, lens_mask, relerr
from glow_tts_amd import attentions
dev = torch.device("cuda:0")
junk = torch.full((64, 1024, 1024), float("nan"), device=dev); del junk   # poison the allocator cache
enc = fill_module(attentions.Encoder(192, 768, 2, 2, 3, 0.1, window_size=4), "enc.").eval()
P = cpu_state(enc, "enc.")
T, lens = 41, [41, 17, 30]
xm = lens_mask(lens, T)
g = torch.Generator().manual_seed(2)
x = torch.randn(3, 192, T, generator=g) * xm
xx = x.clone().requires_grad_(True)
o = R.encoder_fwd(P, "enc.", xx, xm, n_layers=2)
r = torch.randn(o.shape, generator=g)
(o * r).sum().backward()
enc = enc.to(dev)
xd = x.to(dev).requires_grad_(True)
od = enc(xd, xm.to(dev))
(od * r.to(dev)).sum().backward()
print("out", relerr(od.detach().cpu(), o.detach()), "dx", relerr(xd.grad.cpu(), xx.grad))
for name, prm in enc.named_parameters():
    a, b = prm.grad.cpu(), P["enc." + name].grad
    d = (a - b).abs(); idx = d.argmax().item()
    print(f"  {name:32s} rel {relerr(a, b):.4f} nan {torch.isnan(a).sum().item()} maxref {b.abs().max():.3f} worst@{idx}: got {a.flatten()[idx]:.4f} want {b.flatten()[idx]:.4f}")
a, b = enc.ffn_layers[0].conv_1.weight.grad.cpu(), P["enc.ffn_layers.0.conv_1.weight"].grad
e = (a - b).abs().amax(dim=(1, 2))
print("err by co:", [round(v, 2) for v in e.reshape(24, 32).amax(dim=1).tolist()])
e2 = (a - b).abs().amax(dim=(0, 2))
print("err by ci:", [round(v, 2) for v in e2.reshape(6, 32).amax(dim=1).tolist()])
e3 = (a - b).abs().amax(dim=(0, 1))
print("err by tap:", e3.tolist())
ab, bb = enc.ffn_layers[0].conv_1.bias.grad.cpu(), P["enc.ffn_layers.0.conv_1.bias"].grad
print("bias err by 32:", [round(v, 2) for v in (ab - bb).abs().reshape(24, 32).amax(dim=1).tolist()])
