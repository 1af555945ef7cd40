"""Generate tests/golden/float_golden.npz by IMPORTING the reference's Python modules
(/root/reference: modules.py, attentions.py, models.py, commons.py) in the build container.

    make -C oracle && python tests/golden/make_float_golden.py

Nothing of the reference is copied: this script imports it where it lies, fills the reference
nn.Modules with closed-form weights (tests/golden/fill.py), runs them in eval mode on small
seeded inputs at the true channel counts and stores inputs / outputs / input-gradients.

Import notes (recorded in DESIGN.md): commons.py / stft.py import `librosa`, which is not in this
image and is not on the hot path, and models.py imports the in-place-compiled `monotonic_align`
package; both names are registered in sys.modules before the import (librosa as an empty
placeholder whose functions are never called; monotonic_align backed by oracle/_ref, the
reference's own core.pyx compiled by oracle/Makefile).  `text/` is not imported.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GLOWTTS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from fill import fill_module  # noqa: E402
from oracle import mas as omas  # noqa: E402


def import_reference():
    def _never(*a, **k):
        raise RuntimeError("librosa placeholder: not available in this image, not on the hot path")
    lib = types.ModuleType("librosa")
    lib.filters = types.ModuleType("librosa.filters"); lib.filters.mel = _never
    lib.util = types.ModuleType("librosa.util"); lib.util.pad_center = _never; lib.util.tiny = _never
    lib.util.normalize = _never
    lib.stft = _never; lib.istft = _never
    sys.modules.setdefault("librosa", lib)
    sys.modules.setdefault("librosa.filters", lib.filters)
    sys.modules.setdefault("librosa.util", lib.util)

    ma = types.ModuleType("monotonic_align")

    def maximum_path(value, mask):      # call pattern of reference monotonic_align/__init__.py:6-21
        p = omas.oracle_maximum_path(value.detach().cpu().numpy(), mask.detach().cpu().numpy(),
                                     core=omas.ref_maximum_path_c)
        return torch.from_numpy(p).to(device=value.device, dtype=value.dtype)
    ma.maximum_path = maximum_path
    sys.modules.setdefault("monotonic_align", ma)

    sys.path.insert(0, REF)
    import attentions, commons, models, modules  # noqa: E401
    return commons, modules, attentions, models


def lens_mask(lengths, T):
    l = torch.tensor(lengths)
    return (torch.arange(T)[None, :] < l[:, None]).unsqueeze(1).float()


def grads_of(outs_weights, inputs):
    """d(sum_k <out_k, r_k>)/d(inputs) for fixed pseudo-random r_k: pins the backward pass."""
    tot = 0
    for o, seed in outs_weights:
        g = torch.Generator().manual_seed(seed)
        tot = tot + (o * torch.randn(o.shape, generator=g)).sum()
    return torch.autograd.grad(tot, inputs, allow_unused=True)


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    commons, modules, attentions, models = import_reference()
    out = {}
    g = torch.Generator().manual_seed(1234)

    def rnd(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    # ---- ActNorm / InvConvNear on the squeezed flow state [b,160,t]
    x = rnd(2, 160, 12); m = lens_mask([12, 7], 12); x = x * m
    an = fill_module(modules.ActNorm(160), "an.")
    z, ld = an(x, m)
    out.update(an_x=x, an_mask=m, an_z=z, an_logdet=ld)
    ic = fill_module(modules.InvConvNear(160, n_split=4), "ic.")
    z, ld = ic(x, m)
    out.update(ic_z=z, ic_logdet=ld)

    # ---- WN (no conditioning) and WN with g
    xh = rnd(2, 192, 12) * m
    wn = fill_module(modules.WN(160, 192, 5, 1, 4, 0, 0.05), "wn.").eval()
    out.update(wn_x=xh, wn_out=wn(xh, m))
    wng = fill_module(modules.WN(160, 192, 5, 1, 4, 8, 0.05), "wng.").eval()
    gc = rnd(2, 8, 1)
    out.update(wng_g=gc, wng_out=wng(xh, m, gc))

    # ---- CouplingBlock fwd + input grad
    xc = (rnd(2, 160, 12) * m).requires_grad_(True)
    cb = fill_module(attentions.CouplingBlock(160, 192, 5, 1, 4, gin_channels=0, p_dropout=0.05, n_sqz=2), "cb.").eval()
    z, ld = cb(xc, m)
    (gx,) = grads_of([(z, 1), (ld, 2)], [xc])
    out.update(cb_x=xc, cb_z=z, cb_logdet=ld, cb_gx=gx)

    # ---- FlowSpecDecoder (2 blocks), odd T_y to exercise the trim, fwd + input grad
    y = rnd(2, 80, 25); ym = lens_mask([24, 14], 25)
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05, n_split=4, n_sqz=2), "decoder.").eval()
    yy = (y[:, :, :24] * ym[:, :, :24]).requires_grad_(True)
    z, ld = dec(yy, ym[:, :, :24])
    (gy,) = grads_of([(z, 3), (ld, 4)], [yy])
    out.update(dec_y=yy, dec_mask=ym[:, :, :24], dec_z=z, dec_logdet=ld, dec_gy=gy)

    # ---- relative-position MHA at T in {3,5,37}: both branches of attentions.py:292-305
    for T in (3, 5, 37):
        xm = lens_mask([T, max(1, T - 2)], T)
        xa = (rnd(2, 192, T) * xm).requires_grad_(True)
        mha = fill_module(attentions.MultiHeadAttention(192, 192, 2, window_size=4, p_dropout=0.1), f"mha{T}.").eval()
        am = xm.unsqueeze(2) * xm.unsqueeze(-1)
        o = mha(xa, xa, am)
        (ga,) = grads_of([(o, 5)], [xa])
        out.update({f"mha{T}_x": xa, f"mha{T}_mask": xm, f"mha{T}_out": o, f"mha{T}_p": mha.attn, f"mha{T}_gx": ga})

    # ---- FFN, LayerNorm, ConvReluNorm, Encoder(2 layers), TextEncoder(2 layers), DurationPredictor
    T = 11; xm = lens_mask([11, 6], T); xe = rnd(2, 192, T) * xm
    ffn = fill_module(attentions.FFN(192, 192, 768, 3, p_dropout=0.1), "ffn.").eval()
    out.update(enc_x=xe, enc_mask=xm, ffn_out=ffn(xe, xm))
    ln = fill_module(modules.LayerNorm(192), "ln.")
    out.update(ln_out=ln(xe))
    crn = fill_module(modules.ConvReluNorm(192, 192, 192, 5, 3, 0.5), "pre.").eval()
    out.update(crn_out=crn(xe, xm))
    enc = fill_module(attentions.Encoder(192, 768, 2, 2, 3, 0.1, window_size=4), "enc.").eval()
    out.update(encoder_out=enc(xe, xm))
    ids = torch.randint(1, 148, (2, T), generator=g); xl = torch.tensor([11, 6])
    te = fill_module(models.TextEncoder(148, 80, 192, 768, 256, 2, 2, 3, 0.1, window_size=4, mean_only=True,
                                        prenet=True, use_sdp=False), "encoder.").eval()
    tx, tm, tlogs, tmask = te(ids, xl)
    out.update(te_ids=ids, te_len=xl, te_x=tx, te_m=tm, te_logs=tlogs, te_mask=tmask)
    dp = fill_module(models.DurationPredictor(192, 256, 3, 0.1), "dp.").eval()
    out.update(dp_out=dp(xe, xm))

    # ---- glue: logp (models.py:1076-1082), MAS through the reference call pattern, mle_loss
    x_m = rnd(2, 80, T) * xm; x_logs = torch.zeros_like(x_m)
    zz = rnd(2, 80, 24) * ym[:, :, :24]
    with torch.no_grad():
        x_s_sq_r = torch.exp(-2 * x_logs)
        import math
        logp1 = torch.sum(-0.5 * math.log(2 * math.pi) - x_logs, [1]).unsqueeze(-1)
        logp2 = torch.matmul(x_s_sq_r.transpose(1, 2), -0.5 * (zz ** 2))
        logp3 = torch.matmul((x_m * x_s_sq_r).transpose(1, 2), zz)
        logp4 = torch.sum(-0.5 * (x_m ** 2) * x_s_sq_r, [1]).unsqueeze(-1)
        logp = logp1 + logp2 + logp3 + logp4
    import monotonic_align
    amask = (xm.unsqueeze(-1) * ym[:, :, :24].unsqueeze(2)).squeeze(1)
    attn = monotonic_align.maximum_path(logp, amask)
    z_m = torch.matmul(attn.transpose(1, 2), x_m.transpose(1, 2)).transpose(1, 2)
    mle = commons.mle_loss(zz, z_m, torch.zeros_like(z_m), torch.tensor([1.5, -0.5]), ym[:, :, :24])
    out.update(glue_xm=x_m, glue_z=zz, glue_logp=logp, glue_attn=attn, glue_zm=z_m, glue_mle=mle)

    # ---- reverse flow (models.py:765-785 with reverse=True; attentions.py:178-180; modules.py:592-594,647-652) and
    # commons.generate_path (commons.py:127-143) — appended last so that every draw above keeps its value
    zr = rnd(2, 80, 24) * ym[:, :, :24]
    with torch.no_grad():
        xr, ldr = dec(zr, ym[:, :, :24], reverse=True)
    assert ldr is None
    dur = torch.tensor([[3., 1., 0., 2., 4., 1., 0., 0., 0., 0., 0.], [2., 2., 5., 1., 1., 3., 0., 0., 0., 0., 0.]])
    gmask = (xm.unsqueeze(-1) * ym[:, :, :24].unsqueeze(2)).squeeze(1)
    gp = commons.generate_path(dur, gmask)
    out.update(dec_rev_z=zr, dec_rev_x=xr, genpath_dur=dur, genpath_mask=gmask, genpath_out=gp)

    # ---- speaker conditioning g [b,256,1] (cfg 4, configs/base_blank_ms.json): Encoder.cond_g before layer index 2
    # (attentions.py:66-67), DurationPredictor.cond (models.py:587-589), WN.cond_layer inside the decoder
    # (modules.py:148-149) — appended after everything above, so earlier draws keep their values
    spk = rnd(2, 256, 1)
    encg = fill_module(attentions.Encoder(192, 768, 2, 3, 3, 0.1, window_size=4, gin_channels=256), "encg.").eval()
    xeg = xe.clone().requires_grad_(True); spg = spk.clone().requires_grad_(True)
    og = encg(xeg, xm, g=spg)
    gxe, gsp = grads_of([(og, 6)], [xeg, spg])
    out.update(spk_g=spk, encg_out=og, encg_gx=gxe, encg_gg=gsp)
    dpg = fill_module(models.DurationPredictor(192, 256, 3, 0.1, gin_channels=256), "dpg.").eval()
    out.update(dpg_out=dpg(xe, xm, g=spk))
    decg = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05, n_split=4, n_sqz=2, gin_channels=256),
                       "decoder.").eval()
    yg = (y[:, :, :24] * ym[:, :, :24]).requires_grad_(True); spd = spk.clone().requires_grad_(True)
    zg, ldg = decg(yg, ym[:, :, :24], g=spd)
    gyg, gsd = grads_of([(zg, 7), (ldg, 8)], [yg, spd])
    with torch.no_grad():
        xrg, _ = decg(zr, ym[:, :, :24], g=spk, reverse=True)
    out.update(decg_z=zg, decg_logdet=ldg, decg_gy=gyg, decg_gg=gsd, decg_rev_x=xrg)

    # ---- per-frame pitch / energy conditioning (cfg 5): WNP (modules.py:272-362) inside every coupling block
    # (attentions.py:153-154), contours [b,1,t] at the un-squeezed frame rate — appended last again
    pit = rnd(2, 1, 24) * ym[:, :, :24]; ene = rnd(2, 1, 24).abs() * ym[:, :, :24]
    decp = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05, n_split=4, n_sqz=2, gin_channels=256),
                       "decoder.").eval()
    yp = (y[:, :, :24] * ym[:, :, :24]).requires_grad_(True)
    zp, ldp = decp(yp, ym[:, :, :24], g=spk, pitch=pit, energy=ene)
    prm = dict(decp.named_parameters())
    names = ["decoder.flows.2.wn_pitch.cond_layer1.weight_g", "decoder.flows.2.wn_pitch.cond_layer1.weight_v",
             "decoder.flows.2.wn_pitch.cond_layer1.bias", "decoder.flows.5.wn_energy.cond_layer1.weight_v",
             "decoder.flows.5.wn_energy.cond_layer1.bias", "decoder.flows.2.wn_energy.in_layers.1.weight_v"]
    gyp, *gps = grads_of([(zp, 9), (ldp, 10)], [yp] + [prm[n[len("decoder."):]] for n in names])
    with torch.no_grad():
        xrp, _ = decp(zr, ym[:, :, :24], g=spk, pitch=pit, energy=ene, reverse=True)
        zpo, _ = decp(yp, ym[:, :, :24], g=spk, pitch=pit, energy=None)      # pitch only: wn_energy is the identity
    out.update(pros_pitch=pit, pros_energy=ene, decp_z=zp, decp_logdet=ldp, decp_gy=gyp, decp_rev_x=xrp, decp_z_pitch_only=zpo)
    out.update({"decp_g_" + n: v for n, v in zip(names, gps)})

    # ---- CouplingBlock with sigmoid_scale=True (attentions.py:172-173): logs = log(1e-6 + sigmoid(logs + 2))
    cbs = fill_module(attentions.CouplingBlock(160, 192, 5, 1, 4, gin_channels=0, p_dropout=0.05, sigmoid_scale=True, n_sqz=2), "cb.").eval()
    xs_ = xc.detach().clone().requires_grad_(True)
    zs_, lds_ = cbs(xs_, m)
    (gxs_,) = grads_of([(zs_, 11), (lds_, 12)], [xs_])
    out.update(cbs_z=zs_, cbs_logdet=lds_, cbs_gx=gxs_)

    # ---- language vector (cfg 5, lin_channels=4): TextEncoder concatenates it to the (4 channels narrower) token
    # embedding at every position (models.py:654-664, 698-699); DurationPredictor.cond_lang (models.py:582-583, 595-597)
    lng = rnd(2, 4, 1)
    tel = fill_module(models.TextEncoder(148, 80, 192, 768, 256, 2, 2, 3, 0.1, window_size=4, mean_only=True, prenet=True,
                                         use_sdp=False, lin_channels=4), "encoder.").eval()
    ll = lng.clone().requires_grad_(True)
    lx, lm, _, _ = tel(ids, xl, l=ll)
    (gl,) = grads_of([(lx, 13), (lm, 14)], [ll])
    dpl = fill_module(models.DurationPredictor(192, 256, 3, 0.1, gin_channels=256, lin_channels=4), "dpl.").eval()
    out.update(lang_l=lng, tel_x=lx, tel_m=lm, tel_gl=gl, dpl_out=dpl(xe, xm, g=spk, l=lng))

    # ---- ActNorm data-dependent initialisation (modules.py:588-590, 607-619): set_ddi(True) on every ActNorm of a 3-block
    # decoder, one forward — appended after everything above (every section from here on draws from its own generator)
    gd = torch.Generator().manual_seed(4)
    md = lens_mask([44, 30, 6], 44)
    yd = (torch.randn(3, 80, 44, generator=gd) * 1.7 + 0.3) * md
    decd = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 3, 4, p_dropout=0.05, n_split=4, n_sqz=2), "decoder.").eval()
    for f in decd.flows:
        if hasattr(f, "set_ddi"):
            f.set_ddi(True)
    with torch.no_grad():
        zdd, ldd = decd(yd, md)
    out.update(ddi_y=yd, ddi_mask=md, ddi_z=zdd, ddi_logdet=ldd,
               **{f"ddi_logs{b}": decd.flows[3 * b].logs.detach().clone() for b in range(3)},
               **{f"ddi_bias{b}": decd.flows[3 * b].bias.detach().clone() for b in range(3)})

    # ---- DilatedDepthSeparableConv, ElementwiseAffine, ConvFlow (spline), the three stochastic predictors
    import transforms
    gs = torch.Generator().manual_seed(99)
    T = 23; xm = lens_mask([23, 11], T)
    xs_ = (torch.randn(2, 192, T, generator=gs) * xm).requires_grad_(True)
    gcond = torch.randn(2, 192, T, generator=gs) * 0.5
    dds = fill_module(modules.DilatedDepthSeparableConv(192, 3, 3, 0.5), "dds.").eval()
    od = dds(xs_, xm, g=gcond)
    (gxd,) = grads_of([(od, 21)], [xs_])
    out.update(f1_mask=xm, dds_x=xs_, dds_g=gcond, dds_out=od, dds_gx=gxd)
    # the spline itself on inputs that reach both tails and every bin
    sp_in = (torch.randn(2, 1, T, generator=gs) * 3.5).requires_grad_(True)
    uw, uh = torch.randn(2, 1, T, 10, generator=gs).requires_grad_(True), torch.randn(2, 1, T, 10, generator=gs).requires_grad_(True)
    ud = torch.randn(2, 1, T, 9, generator=gs).requires_grad_(True)
    so, sl = transforms.piecewise_rational_quadratic_transform(sp_in, uw, uh, ud, inverse=False, tails="linear", tail_bound=5.0)
    gsp = grads_of([(so, 22), (sl, 23)], [sp_in, uw, uh, ud])
    with torch.no_grad():
        sinv, _ = transforms.piecewise_rational_quadratic_transform(so.detach(), uw, uh, ud, inverse=True, tails="linear", tail_bound=5.0)
    out.update(sp_in=sp_in, sp_uw=uw, sp_uh=uh, sp_ud=ud, sp_out=so, sp_lad=sl, sp_gin=gsp[0], sp_guw=gsp[1], sp_guh=gsp[2],
               sp_gud=gsp[3], sp_inv=sinv)
    ea = fill_module(modules.ElementwiseAffine(2), "ea.")
    z2 = (torch.randn(2, 2, T, generator=gs) * xm)
    eo, el = ea(z2, xm)
    out.update(ea_x=z2, ea_out=eo, ea_logdet=el)
    cf = fill_module(modules.ConvFlow(2, 192, 3, num_layers=3), "cf.").eval()
    z2g = z2.clone().requires_grad_(True); gc2 = gcond.clone().requires_grad_(True)
    co, cl = cf(z2g, xm, g=gc2)
    gz2, gg2 = grads_of([(co, 24), (cl, 25)], [z2g, gc2])
    with torch.no_grad():
        cinv = cf(co.detach(), xm, g=gcond, reverse=True)
    out.update(cf_out=co, cf_logdet=cl, cf_gz=gz2, cf_gg=gg2, cf_inv=cinv)

    class _Noise:                                   # replaces the predictors' torch.randn draws by tensors we keep
        def __init__(self, queue):
            self.q, self.orig = list(queue), torch.randn
        def __enter__(self):
            torch.randn = lambda *a, **k: self.q.pop(0)
        def __exit__(self, *a):
            torch.randn = self.orig
            assert not self.q

    spk5 = torch.randn(2, 512, 1, generator=gs) * 0.5
    lng5 = torch.randn(2, 4, 1, generator=gs)
    xe5 = torch.randn(2, 192, T, generator=gs) * xm
    wdur = (torch.randint(1, 6, (2, 1, T), generator=gs).float()) * xm
    e_w = torch.randn(2, 2, T, generator=gs)
    sdp = fill_module(models.StochasticDurationPredictor(192, 192, 3, 0.5, 4, gin_channels=512, lin_channels=4), "sdp.").eval()
    with _Noise([e_w.clone()]):
        nll_w = sdp(xe5, xm, wdur, g=spk5, l=lng5)
    prm = dict(sdp.named_parameters())
    sdp_names = ["flows.0.log_scale", "flows.2.proj.weight", "flows.4.convs.convs_sep.2.weight", "post_flows.1.pre.weight",
                 "post_flows.3.convs.norms_2.1.gamma", "post_convs.convs_1x1.0.weight", "post_pre.bias", "convs.norms_1.0.beta",
                 "pre.weight", "proj.bias", "cond.weight", "cond_lang.bias", "post_flows.0.translation", "post_proj.weight"]
    gs_ = grads_of([(nll_w, 26)], [prm[n] for n in sdp_names])
    with _Noise([e_w.clone()]):
        logw_rev = sdp(xe5, xm, g=spk5, l=lng5, reverse=True, noise_scale=0.8)
    out.update(p5_g=spk5, p5_l=lng5, p5_x=xe5, p5_w=wdur, p5_ew=e_w, sdp_nll=nll_w, sdp_rev=logw_rev,
               **{"sdp_g_" + n: v for n, v in zip(sdp_names, gs_)})
    Tf = 31; fm = lens_mask([31, 18], Tf)
    xf5 = torch.randn(2, 192, Tf, generator=gs) * fm
    pn5 = torch.randn(2, 1, Tf, generator=gs) * 1.5 * fm
    e_p = torch.randn(2, 1, Tf, generator=gs)
    spp = fill_module(models.StochasticPitchPredictor(192, 256, 3, 0.1, 4, gin_channels=512), "spp.").eval()
    with _Noise([e_p.clone()]):
        nll_p = spp(xf5, fm, pn5, g=spk5)
    prp = dict(spp.named_parameters())
    spp_names = ["flows.0.translation", "flows.1.proj.bias", "flows.3.convs.convs_1x1.2.weight", "convs.convs_sep.1.bias", "pre.bias",
                 "cond.bias", "proj.weight"]
    gp_ = grads_of([(nll_p, 27)], [prp[n] for n in spp_names])
    with _Noise([torch.cat([e_p, e_p.flip(2)], 1).clone()]):
        pit_rev = spp(xf5, fm, g=spk5, reverse=True, noise_scale=0.7)
    sep = fill_module(models.StochasticEnergyPredictor(192, 256, 3, 0.1, 4, gin_channels=512), "sep.").eval()
    with _Noise([e_p.clone()]):
        nll_e = sep(xf5, fm, pn5.abs(), g=spk5)
    out.update(p5_fmask=fm, p5_xf=xf5, p5_pitch=pn5, p5_ep=e_p, spp_nll=nll_p, spp_rev=pit_rev, sep_nll=nll_e,
               **{"spp_g_" + n: v for n, v in zip(spp_names, gp_)})

    # ---- cfg 5 as the reference runs it: the FULL models.FlowGenerator(**hps.model) of configs/base_blank_emo_lang_pitch.json
    # (the only config it constructs for), forward(x, x_lengths, y, y_lengths, g, emo, emo_cartesian, pitch,
    # energy, l) + the training loss of train_ms_emo_lang_pitch.py:295-306 + parameter gradients
    import json
    with open(os.path.join(REF, "configs", "base_blank_emo_lang_pitch.json")) as f:
        hm = json.load(f)["model"]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        gen = models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **hm)
    fill_module(gen, "").eval()
    g5 = torch.Generator().manual_seed(555)
    B5, Tx5, Ty5 = 2, 19, 42
    xl5, yl5 = torch.tensor([19, 12]), torch.tensor([42, 29])
    ids5 = torch.randint(1, 187, (B5, Tx5), generator=g5) * (torch.arange(Tx5)[None, :] < xl5[:, None])
    ym5 = lens_mask(yl5.tolist(), Ty5)
    y5 = torch.randn(B5, 80, Ty5, generator=g5) * ym5
    graw = torch.randn(B5, 512, generator=g5)
    emo5 = torch.tensor([3, 0]); cart5 = torch.tensor([[0.7, 2.1, -0.3], [1.2, 1.7, 2.5]])
    pit5 = ((80 + 200 * torch.rand(B5, 1, Ty5, generator=g5)) * (torch.rand(B5, 1, Ty5, generator=g5) > 0.3)) * ym5
    ene5 = (1 + 10 * torch.rand(B5, 1, Ty5, generator=g5)) * ym5
    lid5 = torch.tensor([2, 0])
    n_w, n_p, n_e = torch.randn(B5, 2, Tx5, generator=g5), torch.randn(B5, 1, Ty5, generator=g5), torch.randn(B5, 1, Ty5, generator=g5)
    with _Noise([n_w.clone(), n_p.clone(), n_e.clone()]):
        (z5, zm5, zlogs5, ld5, zmask5), (xm5, _, xmask5), (attn5, ll5, lp5, le5), _, _ = \
            gen(ids5, xl5, y5, yl5, g=graw, emo=emo5, emo_cartesian=cart5, pitch=pit5, energy=ene5, l=lid5)
    lmle5 = commons.mle_loss(z5, zm5, zlogs5, ld5, zmask5)
    loss5 = lmle5 + torch.sum(ll5.float()) + lp5 * 0.5 + le5 * 0.5
    pg = dict(gen.named_parameters())
    full_names = ["emb_g.weight", "emo_id_proj.weight", "emo_proj.bias", "emo_VAD_inten_proj.weight", "elevation_emb.weight",
                  "azimuth_emb.weight", "sty_proj.weight", "emosty_layer_norm.weight", "emb_l.weight", "encoder.emb.weight",
                  "encoder.encoder.cond_g.weight", "encoder.proj_w.flows.1.proj.weight", "encoder.proj_w.post_flows.2.pre.bias",
                  "proj_pitch.flows.3.proj.weight", "proj_pitch.cond.weight", "proj_energy.flows.0.log_scale",
                  "proj_energy.convs.convs_sep.0.weight", "decoder.flows.2.wn.cond_layer.weight_g",
                  "decoder.flows.35.wn_pitch.cond_layer1.bias", "decoder.flows.17.end.weight", "encoder.proj_m.weight"]
    gfull = torch.autograd.grad(loss5, [pg[n] for n in full_names], allow_unused=True)
    out.update(full_ids=ids5, full_xl=xl5, full_yl=yl5, full_y=y5, full_g=graw, full_emo=emo5, full_cart=cart5, full_pitch=pit5,
               full_energy=ene5, full_lid=lid5, full_nw=n_w, full_np=n_p, full_ne=n_e, full_z=z5, full_zm=zm5, full_logdet=ld5,
               full_attn=attn5, full_l_length=ll5, full_l_pitch=lp5, full_l_energy=le5, full_l_mle=lmle5, full_loss=loss5,
               **{"full_g_" + n: (v if v is not None else torch.zeros_like(pg[n])) for n, v in zip(full_names, gfull)})

    path = os.path.join(HERE, "float_golden.npz")
    np.savez_compressed(path, **{k: v.detach().cpu().numpy() for k, v in out.items()})
    print("wrote", path, len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
