import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests/golden"); sys.path.insert(0, ROOT + "/tests")
import torch
from fill import fill_module
from oracle import glowtts_ref as R
from test_encoder_gpu import cpu_state, lens_mask, relerr
from glow_tts_amd import attentions
dev = torch.device("cuda:0")
for T in (3, 5, 5, 3, 5):
    att = fill_module(attentions.MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=0.1), "mha.").eval()
    P = cpu_state(att, "mha.")
    lens = [T, max(1, T - 2)]
    xm = lens_mask(lens, T)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 192, T, generator=g) * xm
    xx = x.clone().requires_grad_(True)
    am = xm.unsqueeze(2) * xm.unsqueeze(-1)
    o, p = R.mha_fwd(P, "mha.", xx, xx, am)
    r = torch.randn(o.shape, generator=g) * xm
    (o * r).sum().backward()
    att = att.to(dev)
    xd = x.to(dev).requires_grad_(True)
    od = att(xd, xd, am.to(dev))
    (od * r.to(dev)).sum().backward()
    print("T", T, "out", relerr(od.detach().cpu() * xm, o.detach() * xm), "dx", relerr(xd.grad.cpu(), xx.grad))
    for name, prm in att.named_parameters():
        a, b = prm.grad.cpu(), P["mha." + name].grad
        d = (a - b).abs()
        idx = d.argmax().item()
        print(f"  {name:16s} rel {relerr(a, b):.4f} maxref {b.abs().max():.4f} worst at {idx}: got {a.flatten()[idx]:.4f} want {b.flatten()[idx]:.4f}")
