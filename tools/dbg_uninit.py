"""Poison every torch.empty / empty_like with NaN (float) to expose reads of never-written memory (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import train, models
dev = torch.device("cuda:0")
cfg = dict(train.BASE_MODEL, n_blocks_dec=2, n_layers_enc=1, p_dropout=0.0, p_dropout_dec=0.0)
torch.manual_seed(0)
m1 = train.build_model(cfg, device=dev)
with torch.no_grad():
    for n, p in m1.named_parameters():
        if n.endswith("end.weight") or n.endswith("pre.proj.weight"):
            p.normal_(0, 0.02)
batch = train.synth_batch(4, 40, 120, 0, dev)
_empty, _empty_like = torch.empty, torch.empty_like
POISON = float(os.environ.get("POISON", "nan"))
def p_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.device.type == "cuda": t.fill_(POISON)
    return t
def p_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.device.type == "cuda": t.fill_(POISON)
    return t
def run(tag):
    m1.zero_grad(set_to_none=True)
    (z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, _, _), _, _ = m1(*batch)
    l_mle = models.mle_loss(z, z_m, None, logdet, z_mask)
    loss = l_mle + l_length.sum()
    loss.backward()
    bad = [n for n, p in m1.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m1.parameters() if p.grad is not None)).item()
    print(tag, "loss", loss.item(), "mle", l_mle.item(), "len", l_length.sum().item(), "z finite", torch.isfinite(z).all().item(),
          "gnorm", gn, "non-finite grads:", bad[:8], len(bad))
run("clean   ")
torch.empty, torch.empty_like = p_empty, p_empty_like
run("poisoned")
torch.empty, torch.empty_like = _empty, _empty_like
run("clean2  ")
