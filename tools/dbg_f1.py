"""dev: per-parameter gradient error tables of the stochastic predictors, stand-alone and inside FlowGenerator"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import test_predictors_gpu as T
from fill import fill_module
from oracle import glowtts_ref as R
from glow_tts_amd import models, predictors

dev = torch.device("cuda:0")


def table(named, P, prefix, k=12):
    rows = []
    for name, p in named:
        ref = P[prefix + name].grad
        if ref is None or p.grad is None:
            continue
        a, b = p.grad.cpu().double(), ref.double()
        rows.append((float((a - b).norm() / b.norm().clamp_min(1e-12)), float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)),
                     float((a * b).sum() / (b * b).sum().clamp_min(1e-30)), name))
    rows.sort(reverse=True)
    for r in rows[:k]:
        print("   relL2 %.3f  maxabs %.3f  slope %.3f  %s" % r)
    print("   ... median relL2 %.4f over %d params" % (sorted(x[0] for x in rows)[len(rows) // 2], len(rows)))


t = T.t
mod = fill_module(predictors.StochasticDurationPredictor(192, 192, 3, 0.5, 4, gin_channels=512, lin_channels=4), "sdp.").eval()
P = {k: v.requires_grad_(True) for k, v in T.cpu_state(mod, "sdp.").items()}
for wts in ([1.0, -0.7], [0.02, 0.02]):
    for v in P.values():
        v.grad = None
    nll = R.sdp_fwd(P, "sdp.", t("p5_x"), t("f1_mask"), t("p5_w"), t("p5_ew"), g=t("p5_g"), l=t("p5_l"))
    (nll * torch.tensor(wts)).sum().backward()
    m = mod.to(dev)
    for p in m.parameters():
        p.grad = None
    out = m(t("p5_x").to(dev), t("f1_mask").to(dev), t("p5_w").to(dev), noise=t("p5_ew").to(dev), g=t("p5_g").to(dev), l=t("p5_l").to(dev))
    (out * torch.tensor(wts).to(dev)).sum().backward()
    print("stand-alone SDP, weights", wts, "nll", out.tolist(), nll.tolist())
    table(m.named_parameters(), P, "sdp.")

cfg = dict(T.CFG5, n_blocks_dec=2, n_layers_enc=3)
gen = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **cfg), "").eval()
P = {k: v.requires_grad_(v.dtype.is_floating_point and "bins" not in k) for k, v in T.cpu_state(gen).items()}
ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, noise = T._cfg5_inputs(3, 21, 46, seed=9)
gen = gen.to(dev)
d = lambda v: v.to(dev)
(z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, l_pitch, l_energy), _, _ = gen(
    d(ids), d(xl), d(y), d(yl), g=d(graw), emo=d(emo), emo_cartesian=d(cart), pitch=d(pitch), energy=d(energy), l=d(lid), noise=tuple(d(n) for n in noise))
out = R.train_forward_full(P, ids, xl, y, yl, lambda logp, mk: attn.squeeze(1).cpu().float(), cfg, graw, emo, cart, pitch, energy, lid, noise)
for name, a, b in (("l_length", l_length, out["l_length"]), ("l_pitch", l_pitch, out["l_pitch"]), ("l_energy", l_energy, out["l_energy"])):
    print(name, a.detach().cpu().tolist(), b.detach().tolist())
for which, la, lb in (("l_length only", l_length.sum(), out["l_length"].sum()), ("l_pitch only", l_pitch, out["l_pitch"])):
    for p in gen.parameters():
        p.grad = None
    for v in P.values():
        v.grad = None
    la.backward(retain_graph=True); lb.backward(retain_graph=True)
    print("in-model,", which)
    table([(n, p) for n, p in gen.named_parameters() if p.grad is not None and p.requires_grad], P, "", k=14)
# the in-model SDP inputs fed to a stand-alone call
rcx, xb = gen.encoder._last_rows
pw = gen.encoder.proj_w
xin = rcx.from_rows(xb.float())
x_mask = (torch.arange(21, device=dev)[None, :] < d(xl)[:, None]).unsqueeze(1).float()
w = attn.squeeze(1).sum(2).unsqueeze(1)
gv = gen.condition(d(graw), d(emo), d(cart)).detach(); lv = gen.emb_l(d(lid)).unsqueeze(-1).detach()
for p in gen.parameters():
    p.grad = None
nl = pw(xin, x_mask, w, g=gv, l=lv, noise=d(noise[0]))
(nl / x_mask.sum()).sum().backward()
Pq = {k: v.detach().clone().requires_grad_(True) for k, v in P.items() if k.startswith("encoder.proj_w.")}
nq = R.sdp_fwd(Pq, "encoder.proj_w.", xin.cpu(), x_mask.cpu(), w.cpu(), noise[0], g=gv.cpu(), l=lv.cpu())
(nq / x_mask.cpu().sum()).sum().backward()
print("stand-alone call on the in-model inputs: nll", nl.tolist(), nq.tolist())
table(pw.named_parameters(), Pq, "encoder.proj_w.")
