"""CPU tests: the float oracle (oracle/glowtts_ref.py) against goldens produced by IMPORTING the
reference modules (tests/golden/make_float_golden.py).  Weights are the closed-form fill of
tests/golden/fill.py applied by parameter NAME, so these tests also pin the state_dict key/shape
contract of the product's host modules (glow_tts_amd.models / attentions / modules)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from fill import closed_form, filled_state  # noqa: E402
from oracle import glowtts_ref as R  # noqa: E402

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "float_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def close(a, b, tol=2e-5):
    scale = max(1.0, float(b.abs().max()))
    return torch.allclose(a, b, atol=tol * scale, rtol=tol)


def module_state(mod, prefix):
    return {prefix + k: closed_form(prefix + k, v.shape) for k, v in mod.state_dict().items()}


def test_actnorm_invconv():
    P = filled_state({"logs": (1, 160, 1), "bias": (1, 160, 1)}, "an.")
    z, ld = R.actnorm_fwd(P, "an.", t("an_x"), t("an_mask"))
    assert close(z, t("an_z")) and close(ld, t("an_logdet"))
    P = filled_state({"weight": (4, 4)}, "ic.")
    z, ld = R.invconv_fwd(P, "ic.", t("an_x"), t("an_mask"))
    assert close(z, t("ic_z")) and close(ld, t("ic_logdet"))


def test_wn_matches_reference_and_product_state_dict(built):
    from glow_tts_amd import modules
    P = module_state(modules.WN(160, 192, 5, 1, 4, 0, 0.05), "wn.")
    assert close(R.wn_fwd(P, "wn.", t("wn_x"), t("an_mask")), t("wn_out"))
    P = module_state(modules.WN(160, 192, 5, 1, 4, 8, 0.05), "wng.")
    assert close(R.wn_fwd(P, "wng.", t("wn_x"), t("an_mask"), t("wng_g")), t("wng_out"))


def test_coupling_fwd_bwd(built):
    from glow_tts_amd import attentions
    P = module_state(attentions.CouplingBlock(160, 192, 5, 1, 4, p_dropout=0.05), "cb.")
    x = t("cb_x").clone().requires_grad_(True)
    z, ld = R.coupling_fwd(P, "cb.", x, t("an_mask"))
    assert close(z, t("cb_z")) and close(ld, t("cb_logdet"))
    tot = (z * torch.randn(z.shape, generator=torch.Generator().manual_seed(1))).sum() + \
          (ld * torch.randn(ld.shape, generator=torch.Generator().manual_seed(2))).sum()
    (gx,) = torch.autograd.grad(tot, [x])
    assert close(gx, t("cb_gx"), 1e-4)


def test_decoder_fwd_bwd(built):
    from glow_tts_amd import models
    P = module_state(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05), "decoder.")
    y = t("dec_y").clone().requires_grad_(True)
    z, ld = R.decoder_fwd(P, "decoder.", y, t("dec_mask"), n_blocks=2)
    assert close(z, t("dec_z"), 1e-4) and close(ld, t("dec_logdet"), 1e-4)
    tot = (z * torch.randn(z.shape, generator=torch.Generator().manual_seed(3))).sum() + \
          (ld * torch.randn(ld.shape, generator=torch.Generator().manual_seed(4))).sum()
    (gy,) = torch.autograd.grad(tot, [y])
    assert close(gy, t("dec_gy"), 2e-4)


MHA_SHAPES = {"conv_q.weight": (192, 192, 1), "conv_q.bias": (192,), "conv_k.weight": (192, 192, 1), "conv_k.bias": (192,),
              "conv_v.weight": (192, 192, 1), "conv_v.bias": (192,), "conv_o.weight": (192, 192, 1), "conv_o.bias": (192,),
              "emb_rel_k": (1, 9, 96), "emb_rel_v": (1, 9, 96)}
FFN_SHAPES = {"conv_1.weight": (768, 192, 3), "conv_1.bias": (768,), "conv_2.weight": (192, 768, 3), "conv_2.bias": (192,)}


@pytest.mark.parametrize("T", [3, 5, 37])
def test_mha_band_formulation(T):
    """Both branches of attentions.py:292-305 (T < window+1 and T > window+1) equal the 9-diagonal band."""
    P = filled_state(MHA_SHAPES, f"mha{T}.")
    x = t(f"mha{T}_x").clone().requires_grad_(True)
    xm = t(f"mha{T}_mask")
    am = xm.unsqueeze(2) * xm.unsqueeze(-1)
    o, p = R.mha_fwd(P, f"mha{T}.", x, x, am)
    assert close(o, t(f"mha{T}_out")) and close(p, t(f"mha{T}_p"))
    (gx,) = torch.autograd.grad((o * torch.randn(o.shape, generator=torch.Generator().manual_seed(5))).sum(), [x])
    assert close(gx, t(f"mha{T}_gx"), 1e-4)


def enc_shapes(n_layers):
    s = {}
    for i in range(n_layers):
        s.update({f"attn_layers.{i}.{k}": v for k, v in MHA_SHAPES.items()})
        s.update({f"ffn_layers.{i}.{k}": v for k, v in FFN_SHAPES.items()})
        for nm in ("norm_layers_1", "norm_layers_2"):
            s[f"{nm}.{i}.gamma"] = (192,); s[f"{nm}.{i}.beta"] = (192,)
    return s


PRE_SHAPES = {**{f"conv_layers.{i}.weight": (192, 192, 5) for i in range(3)}, **{f"conv_layers.{i}.bias": (192,) for i in range(3)},
              **{f"norm_layers.{i}.gamma": (192,) for i in range(3)}, **{f"norm_layers.{i}.beta": (192,) for i in range(3)},
              "proj.weight": (192, 192, 1), "proj.bias": (192,)}
DP_SHAPES = {"conv_1.weight": (256, 192, 3), "conv_1.bias": (256,), "norm_1.gamma": (256,), "norm_1.beta": (256,),
             "conv_2.weight": (256, 256, 3), "conv_2.bias": (256,), "norm_2.gamma": (256,), "norm_2.beta": (256,),
             "proj.weight": (1, 256, 1), "proj.bias": (1,)}


def test_encoder_pieces():
    x, m = t("enc_x"), t("enc_mask")
    assert close(R.ffn_fwd(filled_state(FFN_SHAPES, "ffn."), "ffn.", x, m), t("ffn_out"))
    P = filled_state({"gamma": (192,), "beta": (192,)}, "ln.")
    assert close(R.layer_norm_c(x, P["ln.gamma"], P["ln.beta"]), t("ln_out"))
    assert close(R.conv_relu_norm_fwd(filled_state(PRE_SHAPES, "pre."), "pre.", x, m), t("crn_out"))
    assert close(R.encoder_fwd(filled_state(enc_shapes(2), "enc."), "enc.", x, m, n_layers=2), t("encoder_out"), 1e-4)
    assert close(R.duration_predictor_fwd(filled_state(DP_SHAPES, "dp."), "dp.", x, m), t("dp_out"), 1e-4)


def test_text_encoder():
    shapes = {"emb.weight": (148, 192), "proj_m.weight": (80, 192, 1), "proj_m.bias": (80,)}
    shapes.update({"pre." + k: v for k, v in PRE_SHAPES.items()})
    shapes.update({"encoder." + k: v for k, v in enc_shapes(2).items()})
    shapes.update({"proj_w." + k: v for k, v in DP_SHAPES.items()})
    P = filled_state(shapes, "encoder.")
    x, xm_, xl, mask = R.text_encoder_fwd(P, "encoder.", torch.from_numpy(G["te_ids"]), torch.from_numpy(G["te_len"]), n_layers=2)
    assert close(x, t("te_x"), 1e-4) and close(xm_, t("te_m"), 1e-4) and torch.equal(mask, t("te_mask"))


def test_glue_logp_mas_mle(built):
    from oracle import mas as omas
    logp = R.logp_lattice(t("glue_xm"), torch.zeros_like(t("glue_xm")), t("glue_z"))
    assert close(logp, t("glue_logp"), 1e-5)
    xm, ym = t("enc_mask"), t("dec_mask")
    amask = (xm.unsqueeze(-1) * ym.unsqueeze(2)).squeeze(1)
    attn = torch.from_numpy(omas.oracle_maximum_path(t("glue_logp").numpy(), amask.numpy())).float()
    assert torch.equal(attn, t("glue_attn"))
    z_m = torch.matmul(attn.transpose(1, 2), t("glue_xm").transpose(1, 2)).transpose(1, 2)
    assert close(z_m, t("glue_zm"))
    mle = R.mle_loss(t("glue_z"), z_m, torch.zeros_like(z_m), torch.tensor([1.5, -0.5]), ym)
    assert close(mle, t("glue_mle"))


def test_reverse_flow_and_generate_path_match_reference_golden(built):
    """Inference direction: FlowSpecDecoder(reverse=True) and commons.generate_path of the imported reference
    (fixtures appended to float_golden.npz by make_float_golden.py) against the oracle restatement; and the restated
    reverse undoes the restated forward."""
    from glow_tts_amd import models
    P = module_state(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05), "decoder.")
    z, mask = t("dec_rev_z"), t("dec_mask")
    x = R.decoder_rev(P, "decoder.", z, mask, n_blocks=2)
    assert close(x, t("dec_rev_x"), 2e-4), (x - t("dec_rev_x")).abs().max()
    zz, _ = R.decoder_fwd(P, "decoder.", x, mask, n_blocks=2)
    assert close(zz, z, 1e-3)
    gp = R.generate_path(t("genpath_dur"), t("genpath_mask"))
    assert torch.equal(gp, t("genpath_out"))


def test_speaker_conditioning_matches_reference_golden(built):
    """cfg 4 (configs/base_blank_ms.json, gin_channels=256): the speaker vector g [b,256,1] through Encoder.cond_g
    (attentions.py:66-67), DurationPredictor.cond (models.py:587-589) and the WN cond_layer of every coupling block
    (modules.py:148-149; forward, both gradients, reverse) — oracle restatement vs the imported reference."""
    from glow_tts_amd import models
    x, m, spk = t("enc_x"), t("enc_mask"), t("spk_g")
    sh = dict(enc_shapes(3)); sh.update({"cond_g.weight": (192, 256), "cond_g.bias": (192,)})
    xx, gg = x.clone().requires_grad_(True), spk.clone().requires_grad_(True)
    o = R.encoder_fwd(filled_state(sh, "encg."), "encg.", xx, m, g=gg, n_layers=3)
    assert close(o, t("encg_out"), 1e-4)
    gx, g_g = torch.autograd.grad((o * torch.randn(o.shape, generator=torch.Generator().manual_seed(6))).sum(), [xx, gg])
    assert close(gx, t("encg_gx"), 2e-4) and close(g_g, t("encg_gg"), 2e-4)
    sh = dict(DP_SHAPES); sh.update({"cond.weight": (192, 256, 1), "cond.bias": (192,)})
    assert close(R.duration_predictor_fwd(filled_state(sh, "dpg."), "dpg.", x, m, g=spk), t("dpg_out"), 1e-4)
    P = module_state(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05, gin_channels=256), "decoder.")
    y, gg = t("dec_y").clone().requires_grad_(True), spk.clone().requires_grad_(True)
    z, ld = R.decoder_fwd(P, "decoder.", y, t("dec_mask"), gg, n_blocks=2)
    assert close(z, t("decg_z"), 1e-4) and close(ld, t("decg_logdet"), 1e-4)
    tot = (z * torch.randn(z.shape, generator=torch.Generator().manual_seed(7))).sum() + \
          (ld * torch.randn(ld.shape, generator=torch.Generator().manual_seed(8))).sum()
    gy, g_g = torch.autograd.grad(tot, [y, gg])
    assert close(gy, t("decg_gy"), 2e-4) and close(g_g, t("decg_gg"), 2e-4)
    xr = R.decoder_rev(P, "decoder.", t("dec_rev_z"), t("dec_mask"), spk, n_blocks=2)
    assert close(xr, t("decg_rev_x"), 2e-4)


PROS_GRADS = ["decoder.flows.2.wn_pitch.cond_layer1.weight_g", "decoder.flows.2.wn_pitch.cond_layer1.weight_v",
              "decoder.flows.2.wn_pitch.cond_layer1.bias", "decoder.flows.5.wn_energy.cond_layer1.weight_v",
              "decoder.flows.5.wn_energy.cond_layer1.bias", "decoder.flows.2.wn_energy.in_layers.1.weight_v"]


def test_pitch_energy_conditioning_matches_reference_golden(built):
    """cfg 5's per-frame conditioning: modules.WNP (modules.py:272-362) as wn_energy / wn_pitch of every coupling block
    (attentions.py:153-154), next to the speaker vector — forward, log-det, input gradient, parameter gradients of the
    cond_layer1 affine maps and of a WNP conv, reverse; pitch alone (wn_energy the identity).  Oracle restatement vs the
    imported reference; the state_dict comes from the product's module (key / shape contract)."""
    from glow_tts_amd import models
    dec = models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05, gin_channels=256, with_prosody_wn=True)
    P = module_state(dec, "decoder.")
    for n in PROS_GRADS:
        assert n in P, n
        P[n].requires_grad_(True)
    spk, pit, ene, mask = t("spk_g"), t("pros_pitch"), t("pros_energy"), t("dec_mask")
    y = t("dec_y").clone().requires_grad_(True)
    z, ld = R.decoder_fwd(P, "decoder.", y, mask, spk, n_blocks=2, pitch=pit, energy=ene)
    assert close(z, t("decp_z"), 1e-4) and close(ld, t("decp_logdet"), 1e-4)
    tot = (z * torch.randn(z.shape, generator=torch.Generator().manual_seed(9))).sum() + \
          (ld * torch.randn(ld.shape, generator=torch.Generator().manual_seed(10))).sum()
    gy, *gp = torch.autograd.grad(tot, [y] + [P[n] for n in PROS_GRADS])
    assert close(gy, t("decp_gy"), 2e-4)
    for n, g_ in zip(PROS_GRADS, gp):
        assert close(g_, t("decp_g_" + n), 5e-4), n
    with torch.no_grad():
        assert close(R.decoder_rev(P, "decoder.", t("dec_rev_z"), mask, spk, n_blocks=2, pitch=pit, energy=ene), t("decp_rev_x"), 2e-4)
        zo, _ = R.decoder_fwd(P, "decoder.", y, mask, spk, n_blocks=2, pitch=pit, energy=None)
        assert close(zo, t("decp_z_pitch_only"), 1e-4)


def test_coupling_sigmoid_scale_matches_reference_golden(built):
    """sigmoid_scale=True (attentions.py:172-173): logs = log(1e-6 + sigmoid(logs + 2)) — forward, log-det, input gradient."""
    from glow_tts_amd import attentions
    P = module_state(attentions.CouplingBlock(160, 192, 5, 1, 4, p_dropout=0.05, sigmoid_scale=True), "cb.")
    x = t("cb_x").clone().requires_grad_(True)
    z, ld = R.coupling_fwd(P, "cb.", x, t("an_mask"), sigmoid_scale=True)
    assert close(z, t("cbs_z")) and close(ld, t("cbs_logdet"))
    tot = (z * torch.randn(z.shape, generator=torch.Generator().manual_seed(11))).sum() + \
          (ld * torch.randn(ld.shape, generator=torch.Generator().manual_seed(12))).sum()
    (gx,) = torch.autograd.grad(tot, [x])
    assert close(gx, t("cbs_gx"), 1e-4)


def test_language_vector_matches_reference_golden():
    """cfg 5's multi-language inputs: TextEncoder(lin_channels=4) concatenates the language vector to the narrower token
    embedding (models.py:654-664, 698-699) and DurationPredictor adds cond_lang(l) (models.py:582-583, 595-597)."""
    shapes = {"emb.weight": (148, 188), "proj_m.weight": (80, 192, 1), "proj_m.bias": (80,)}
    shapes.update({"pre." + k: v for k, v in PRE_SHAPES.items()})
    shapes.update({"encoder." + k: v for k, v in enc_shapes(2).items()})
    dps = dict(DP_SHAPES); dps.update({"cond_lang.weight": (192, 4, 1), "cond_lang.bias": (192,)})
    shapes.update({"proj_w." + k: v for k, v in dps.items()})
    P = filled_state(shapes, "encoder.")
    l = t("lang_l").clone().requires_grad_(True)
    x, xm_, _, _ = R.text_encoder_fwd(P, "encoder.", torch.from_numpy(G["te_ids"]), torch.from_numpy(G["te_len"]), n_layers=2, l=l)
    assert close(x, t("tel_x"), 1e-4) and close(xm_, t("tel_m"), 1e-4)
    tot = (x * torch.randn(x.shape, generator=torch.Generator().manual_seed(13))).sum() + \
          (xm_ * torch.randn(xm_.shape, generator=torch.Generator().manual_seed(14))).sum()
    (gl,) = torch.autograd.grad(tot, [l])
    assert close(gl, t("tel_gl"), 2e-4)
    sh = dict(DP_SHAPES); sh.update({"cond.weight": (192, 256, 1), "cond.bias": (192,), "cond_lang.weight": (192, 4, 1), "cond_lang.bias": (192,)})
    out = R.duration_predictor_fwd(filled_state(sh, "dpl."), "dpl.", t("enc_x"), t("enc_mask"), g=t("spk_g"), l=t("lang_l"))
    assert close(out, t("dpl_out"), 1e-4)


# ------------------------------------------------------------------------------------------------ round 2: DDI, f1, cfg 5
def _shapes_state(shapes, prefix):
    return {prefix + k: closed_form(prefix + k, v) for k, v in shapes.items()}


def test_actnorm_ddi(built):
    """ActNorm.initialize (modules.py:607-619) block after block, against the reference decoder run with set_ddi(True)."""
    from glow_tts_amd import models
    P = module_state(models.FlowSpecDecoder(80, 192, 5, 1, 3, 4, p_dropout=0.05), "decoder.")
    P2, z, ld = R.decoder_ddi(P, "decoder.", t("ddi_y"), t("ddi_mask"), n_blocks=3)
    for b in range(3):
        assert close(P2[f"decoder.flows.{3 * b}.logs"], t(f"ddi_logs{b}"), 1e-4), b
        assert close(P2[f"decoder.flows.{3 * b}.bias"], t(f"ddi_bias{b}"), 1e-4), b
    assert close(z, t("ddi_z"), 1e-4) and close(ld, t("ddi_logdet"), 1e-4)


def test_spline_matches_reference_transform():
    """transforms.piecewise_rational_quadratic_transform (linear tails, 10 bins, bound 5): outputs, log|det|, all four
    input gradients, and the inverse branch."""
    x, uw, uh, ud = (t(k).clone().requires_grad_(True) for k in ("sp_in", "sp_uw", "sp_uh", "sp_ud"))
    assert (t("sp_in").abs() > 5).any() and (t("sp_in").abs() < 5).any()
    o, lad = R.rq_spline_fwd(x, uw, uh, ud)
    assert close(o, t("sp_out")) and close(lad, t("sp_lad"))
    tot = (o * torch.randn(o.shape, generator=torch.Generator().manual_seed(22))).sum() + \
          (lad * torch.randn(lad.shape, generator=torch.Generator().manual_seed(23))).sum()
    gs = torch.autograd.grad(tot, [x, uw, uh, ud])
    for got, name in zip(gs, ("sp_gin", "sp_guw", "sp_guh", "sp_gud")):
        assert close(got, t(name), 1e-4), name
    assert close(R.rq_spline_inv(t("sp_out"), t("sp_uw"), t("sp_uh"), t("sp_ud")), t("sp_inv"), 1e-4)


def test_dds_conv_elementwise_affine_conv_flow(built):
    from glow_tts_amd import predictors
    P = module_state(predictors.DilatedDepthSeparableConv(192, 3, 3, 0.5), "dds.")
    x = t("dds_x").clone().requires_grad_(True)
    o = R.dds_conv(P, "dds.", x, t("f1_mask"), g=t("dds_g"))
    assert close(o, t("dds_out"), 1e-4)
    (gx,) = torch.autograd.grad((o * torch.randn(o.shape, generator=torch.Generator().manual_seed(21))).sum(), [x])
    assert close(gx, t("dds_gx"), 1e-4)
    P = module_state(predictors.ElementwiseAffine(2), "ea.")
    y, ld = R.elementwise_affine(P, "ea.", t("ea_x"), t("f1_mask"))
    assert close(y, t("ea_out")) and close(ld, t("ea_logdet"))
    P = module_state(predictors.ConvFlow(2, 192, 3, num_layers=3), "cf.")
    z, g = t("ea_x").clone().requires_grad_(True), t("dds_g").clone().requires_grad_(True)
    co, cl = R.conv_flow(P, "cf.", z, t("f1_mask"), g)
    assert close(co, t("cf_out"), 1e-4) and close(cl, t("cf_logdet"), 1e-4)
    tot = (co * torch.randn(co.shape, generator=torch.Generator().manual_seed(24))).sum() + \
          (cl * torch.randn(cl.shape, generator=torch.Generator().manual_seed(25))).sum()
    gz, gg = torch.autograd.grad(tot, [z, g])
    assert close(gz, t("cf_gz"), 2e-4) and close(gg, t("cf_gg"), 2e-4)
    assert close(R.conv_flow(P, "cf.", t("cf_out"), t("f1_mask"), t("dds_g"), reverse=True), t("cf_inv"), 2e-4)


def _grad_names(prefix):
    return [k[len(prefix):] for k in G.files if k.startswith(prefix)]


def test_stochastic_predictors(built):
    """StochasticDurationPredictor / StochasticPitchPredictor / StochasticEnergyPredictor (models.py:217-481): nll with the
    noise draws injected, parameter gradients, and the reverse (synthesis) branch."""
    from glow_tts_amd import predictors
    P = {k: v.requires_grad_(True) for k, v in
         module_state(predictors.StochasticDurationPredictor(192, 192, 3, 0.5, 4, gin_channels=512, lin_channels=4), "sdp.").items()}
    nll = R.sdp_fwd(P, "sdp.", t("p5_x"), t("f1_mask"), t("p5_w"), t("p5_ew"), g=t("p5_g"), l=t("p5_l"))
    assert close(nll, t("sdp_nll"), 1e-4), (nll, t("sdp_nll"))
    names = _grad_names("sdp_g_")
    assert len(names) >= 10
    gs = torch.autograd.grad((nll * torch.randn(nll.shape, generator=torch.Generator().manual_seed(26))).sum(), [P["sdp." + n] for n in names])
    for n, got in zip(names, gs):
        assert close(got, t("sdp_g_" + n), 5e-4), n
    Pd = {k: v.detach() for k, v in P.items()}
    rev = R.predictor_reverse(Pd, "sdp.", t("p5_x"), t("f1_mask"), t("p5_ew") * 0.8, g=t("p5_g"), l=t("p5_l"))
    assert close(rev, t("sdp_rev"), 5e-4)
    P = {k: v.requires_grad_(True) for k, v in
         module_state(predictors.StochasticPitchPredictor(192, 256, 3, 0.1, 4, gin_channels=512), "spp.").items()}
    nll = R.spp_fwd(P, "spp.", t("p5_xf"), t("p5_fmask"), t("p5_pitch"), t("p5_ep"), g=t("p5_g"))
    assert close(nll, t("spp_nll"), 1e-4)
    names = _grad_names("spp_g_")
    gs = torch.autograd.grad((nll * torch.randn(nll.shape, generator=torch.Generator().manual_seed(27))).sum(), [P["spp." + n] for n in names])
    for n, got in zip(names, gs):
        assert close(got, t("spp_g_" + n), 5e-4), n
    Pd = {k: v.detach() for k, v in P.items()}
    noise = torch.cat([t("p5_ep"), t("p5_ep").flip(2)], 1) * 0.7
    assert close(R.predictor_reverse(Pd, "spp.", t("p5_xf"), t("p5_fmask"), noise, g=t("p5_g")), t("spp_rev"), 5e-4)
    P = module_state(predictors.StochasticEnergyPredictor(192, 256, 3, 0.1, 4, gin_channels=512), "sep.")
    nll = R.spp_fwd(P, "sep.", t("p5_xf"), t("p5_fmask"), t("p5_pitch").abs(), t("p5_ep"), g=t("p5_g"))
    assert close(nll, t("sep_nll"), 1e-4)


CFG5 = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1, n_blocks_dec=12,
            n_layers_enc=10, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5, n_block_layers=4, n_sqz=2,
            prenet=True, mean_only=True, hidden_channels_enc=192, hidden_channels_dec=192, window_size=4, gin_channels=512,
            use_sdp=True, use_spk_embeds=True, use_lang_embeds=True, use_emo_embeds=True, lin_channels=4, emoin_channels=1024,
            use_spp=True, use_sep=True)      # == configs/base_blank_emo_lang_pitch.json "model" (the keys FlowGenerator reads)


def test_full_cfg5_flow_generator(built):
    """configs/base_blank_emo_lang_pitch.json as the reference runs it: the oracle's train_forward_full against the reference's
    own FlowGenerator.forward (speaker / emotion front end, SDP / SPP / SEP losses, 12 blocks x 3 WaveNets) — every output of
    the 5-tuple the training loop uses, the loss, and 21 parameter gradients.  Also pins the product's state_dict."""
    from glow_tts_amd import models
    from oracle import mas as omas
    gen = models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **CFG5)
    P = {k: closed_form(k, v.shape) for k, v in gen.state_dict().items() if v.dtype.is_floating_point}
    P["elevation_bins"], P["azimuth_bins"] = gen.elevation_bins.detach().clone(), gen.azimuth_bins.detach().clone()
    for v in P.values():
        v.requires_grad_(True)

    def mp(logp, mask):
        return torch.from_numpy(omas.oracle_maximum_path(logp.numpy(), mask.numpy())).float()
    out = R.train_forward_full(P, t("full_ids"), t("full_xl"), t("full_y"), t("full_yl"), mp, CFG5, t("full_g"), t("full_emo"),
                               t("full_cart"), t("full_pitch"), t("full_energy"), t("full_lid"), (t("full_nw"), t("full_np"), t("full_ne")))
    assert torch.equal(out["attn"], t("full_attn"))
    for key, name, tol in (("z", "full_z", 2e-4), ("z_m", "full_zm", 2e-4), ("logdet", "full_logdet", 2e-4),
                           ("l_length", "full_l_length", 2e-4), ("l_pitch", "full_l_pitch", 2e-4), ("l_energy", "full_l_energy", 2e-4),
                           ("l_mle", "full_l_mle", 2e-4), ("loss", "full_loss", 2e-4)):
        assert close(out[key], t(name), tol), (key, out[key], t(name))
    names = _grad_names("full_g_")
    assert len(names) == 21
    gs = torch.autograd.grad(out["loss"], [P[n] for n in names], allow_unused=True)
    for n, got in zip(names, gs):
        got = torch.zeros_like(P[n]) if got is None else got
        assert close(got, t("full_g_" + n), 2e-3), (n, (got - t("full_g_" + n)).abs().max().item(), t("full_g_" + n).abs().max().item())
