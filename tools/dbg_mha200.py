import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
faulthandler.enable()
import torch
from glow_tts_amd import attentions
from fill import closed_form
dev = torch.device("cuda:0")
att = attentions.MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=0.1)
with torch.no_grad():
    for n, p in att.named_parameters():
        p.copy_(closed_form("mha." + n, p.shape))
att = att.eval().to(dev)
for it in range(6):
    for T in (200, 150, 256, 161, 300, 375, 257, 384):
        lens = [T, max(1, T - 2)]
        xm = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).unsqueeze(1).float()
        x = (torch.randn(2, 192, T) * xm).to(dev).requires_grad_(True)
        am = (xm.unsqueeze(2) * xm.unsqueeze(-1)).to(dev)
        print("iter", it, "T", T, "fwd", flush=True)
        o = att(x, x, am)
        torch.cuda.synchronize()
        print("iter", it, "T", T, "bwd", flush=True)
        (o * torch.randn_like(o)).sum().backward()
        torch.cuda.synchronize()
print("done")
