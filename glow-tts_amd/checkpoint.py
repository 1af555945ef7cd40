"""Checkpoint wire format of the reference (utils.py:85-131): one `torch.save` dict
    {'model': state_dict, 'iteration': int, 'optimizer': AdamW.state_dict(), 'scheduler': ..., 'learning_rate': float}
with the reference's parameter names (weight-norm convs as weight_g / weight_v), so that a `G_*.pth` written by the
reference's training script loads here and one written here loads there.  The flat AdamW of train.FlatAdamW keeps its
moments in two flat buffers; on the wire they are torch.optim.AdamW's per-parameter `exp_avg` / `exp_avg_sq` / `step`
in the order of `model.parameters()` (the order the reference hands to `torch.optim.AdamW`,
train_ms_emo_lang_pitch.py:160).  Files are read with `weights_only=True`: nothing from the file is executed.

  save_checkpoint / load_checkpoint   utils.py:117-131 / 85-115 (parameters missing from the file keep their value)
  warm_start_model                    utils.py:18-83 with transfer_weight (utils.py:366-384): tensors that grew are
                                      padded with N(0,1) entries, `ignore_layers` and still-mismatched ones are skipped
"""
import os

import torch


def _model_order(trainer):
    """index of every trainable parameter of the model in the flat layout, in model.parameters() order"""
    pos = {id(p): i for i, p in enumerate(trainer.buckets.params)}
    return [pos[id(p)] for p in trainer.model.parameters() if p.requires_grad]


def optimizer_state_dict(trainer):
    """torch.optim.AdamW.state_dict() layout of the flat optimizer state."""
    opt, gb = trainer.opt, trainer.buckets
    lr, b1, b2, eps, wd, step = [float(v) for v in opt.hyper.tolist()]
    state = {}
    for k, i in enumerate(_model_order(trainer)):
        o, n, shape = gb.offsets[i], gb.params[i].numel(), gb.params[i].shape
        state[k] = {"step": torch.tensor(step), "exp_avg": opt.m[o:o + n].view(shape).detach().cpu().clone(),
                    "exp_avg_sq": opt.v[o:o + n].view(shape).detach().cpu().clone()}
    group = {"lr": lr, "betas": (b1, b2), "eps": eps, "weight_decay": wd, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(state)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd):
    opt, gb = trainer.opt, trainer.buckets
    order = _model_order(trainer)
    groups = sd["param_groups"]
    idx = [k for g in groups for k in g["params"]]
    if len(idx) != len(order):
        raise ValueError(f"optimizer state holds {len(idx)} parameters, the model has {len(order)}")
    step = 0.0
    with torch.no_grad():
        for k, i in zip(idx, order):
            st = sd["state"].get(k)
            if st is None:
                continue
            o, n = gb.offsets[i], gb.params[i].numel()
            opt.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            opt.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            step = max(step, float(st["step"]))
        g0 = groups[0]
        opt.hyper.copy_(torch.tensor([g0["lr"], g0["betas"][0], g0["betas"][1], g0["eps"], g0.get("weight_decay", 0.01), step]))


def save_checkpoint(trainer, learning_rate, iteration, checkpoint_path):
    """reference utils.save_checkpoint (utils.py:117-131)"""
    torch.save({"model": {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items()},
                "iteration": int(iteration), "optimizer": optimizer_state_dict(trainer),
                "scheduler": {"last_epoch": int(trainer.n_steps), "total_steps": int(trainer.total_steps or 0)},
                "learning_rate": float(learning_rate)}, checkpoint_path)


def load_checkpoint(checkpoint_path, model, trainer=None):
    """reference utils.load_checkpoint (utils.py:85-115): -> (learning_rate, iteration).  Parameters the file does not
    hold keep their current value.  With a trainer, the optimizer moments / step and the schedule position are restored."""
    assert os.path.isfile(checkpoint_path)
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    saved = ck["model"]
    cur = model.state_dict()
    # the reference keeps the model's value for every key the file lacks and says so (utils.py:103-106: "%s is not in the
    # checkpoint"); keys the FILE holds and this model does not (a cfg 5 checkpoint into a base-config model, say) are reported
    # too instead of being dropped silently — load_checkpoint.last_report = {"missing": [...], "ignored": [...]}
    missing = sorted(k for k in cur if k not in saved)
    ignored = sorted(k for k in saved if k not in cur)
    load_checkpoint.last_report = {"missing": missing, "ignored": ignored}
    if missing or ignored:
        import warnings
        warnings.warn(f"load_checkpoint({os.path.basename(checkpoint_path)}): {len(missing)} model parameters are not in the checkpoint "
                      f"(kept: {missing[:4]}{' ...' if len(missing) > 4 else ''}); {len(ignored)} checkpoint entries have no counterpart "
                      f"in this model (ignored: {ignored[:4]}{' ...' if len(ignored) > 4 else ''})")
    model.load_state_dict({k: saved.get(k, v) for k, v in cur.items()})
    if trainer is not None:
        trainer.invalidate_packed()                  # (load_state_dict bumps the parameters' version counters anyway)
        if "optimizer" in ck:
            load_optimizer_state_dict(trainer, ck["optimizer"])
        if "scheduler" in ck and "last_epoch" in ck["scheduler"]:
            trainer.n_steps = int(ck["scheduler"]["last_epoch"])
    return ck.get("learning_rate"), ck.get("iteration", 1)


def transfer_weight(original, target_size, generator=None):
    """reference utils.transfer_weight (utils.py:366-384): pad every dimension that grew with N(0,1) entries."""
    for i, want in enumerate(target_size):
        diff = want - original.size(i)
        if diff > 0:
            dims = list(original.size())
            dims[i] = diff
            original = torch.cat([original, torch.randn(*dims, generator=generator)], dim=i)
    return original


def warm_start_model(checkpoint_path, model, ignore_layers=(), generator=None):
    """reference utils.warm_start_model (utils.py:18-83): load what fits, grow what grew, skip `ignore_layers` and
    tensors whose shape still differs.  Returns (model, grown keys, skipped keys)."""
    assert os.path.isfile(checkpoint_path)
    saved = dict(torch.load(checkpoint_path, map_location="cpu", weights_only=True)["model"])
    cur = model.state_dict()
    grown, mismatched = [], []
    for k, v in list(saved.items()):
        if k in cur and v.size() != cur[k].size():
            try:
                saved[k] = transfer_weight(v, cur[k].size(), generator)
                (grown if saved[k].size() == cur[k].size() else mismatched).append(k)
            except Exception:
                mismatched.append(k)
    skip = set(ignore_layers) | set(mismatched)
    new = dict(cur)
    new.update({k: v for k, v in saved.items() if k not in skip and k in cur})
    model.load_state_dict(new, strict=False)
    return model, grown, sorted(skip)
