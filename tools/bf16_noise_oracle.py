"""dev (CPU only): how far do the ORACLE's own parameter gradients move when every GEMM operand is rounded to bf16 (what the
MFMA path does)?  Separates rounding noise from kernel bugs when a parity test reports a large relative error on a small
gradient: cfg 5 at the test shape gives 0.47 on emb_l.weight (a per-utterance sum of ~1e-3 entries) and > 1 on the
stochastic predictors' parameters — which is why their 192x192 products run as bf16x3 split GEMMs."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, torch.nn.functional as F
import test_predictors_gpu as T
from fill import fill_module
from oracle import glowtts_ref as R, mas as omas
from glow_tts_amd import models
cfg = dict(T.CFG5, n_blocks_dec=2, n_layers_enc=3)
gen = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **cfg), "").eval()
ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, noise = T._cfg5_inputs(3, 21, 46, seed=9)
mp = lambda logp, mask: torch.from_numpy(omas.oracle_maximum_path(logp.numpy(), mask.numpy())).float()
orig = F.conv1d
def bf(x): return x.to(torch.bfloat16).float()
def conv_bf(x, w, b=None, *a, **k):
    if k.get("groups", 1) == 1 and w.shape[1] > 1:
        return orig(bf(x), bf(w), b, *a, **k)
    return orig(x, w, b, *a, **k)
res = []
attn = None
for conv in (orig, conv_bf):
    R.F.conv1d = conv
    P = {k: v.requires_grad_(v.dtype.is_floating_point and "bins" not in k) for k, v in T.cpu_state(gen).items()}
    if attn is None:
        out = R.train_forward_full(P, ids, xl, y, yl, mp, cfg, graw, emo, cart, pitch, energy, lid, noise)
        attn = out["attn"].squeeze(1).detach()
    else:
        out = R.train_forward_full(P, ids, xl, y, yl, lambda a, b: attn, cfg, graw, emo, cart, pitch, energy, lid, noise)
    out["loss"].backward()
    res.append({k: v.grad for k, v in P.items() if v.grad is not None})
R.F.conv1d = orig
rows = []
for k in res[0]:
    a, b = res[1][k].double(), res[0][k].double()
    if b.norm() < 1e-9: continue
    rows.append((float((a - b).norm() / b.norm()), k))
rows.sort(reverse=True)
for r in rows[:12]: print("%.3f %s" % r)
print("emb_l:", [r for r in rows if r[1] == "emb_l.weight"], "lid", lid.tolist())
print(res[0]["emb_l.weight"][:3])
