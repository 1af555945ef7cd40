"""GPU parity tests of SURVEY §8 f1 and the cfg-5 drop-in: DilatedDepthSeparableConv, ConvFlow + rational-quadratic
spline, the stochastic duration / pitch / energy predictors (HIP kernels of csrc/predictor_ops.hip through the C-ABI)
and the full FlowGenerator of configs/base_blank_emo_lang_pitch.json, against the float oracle (oracle/glowtts_ref.py,
pinned to the reference's own modules by tests/golden/float_golden.npz: dds_*, sp_*, cf_*, sdp_*, spp_*, sep_*, full_*).

Tolerances: the 192x192 1x1 convs run in bf16 with fp32 accumulation, everything else (depthwise convs, LayerNorms, GELU,
the 29-row proj, the spline, the likelihood sums) in fp32: activations 2e-2 of max-abs, nll 1e-2 relative, parameter
gradients 8e-2 of max-abs (bf16 operands of the weight-gradient GEMMs)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from fill import fill_module  # noqa: E402
from oracle import glowtts_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "float_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


def cpu_state(mod, prefix=""):
    return {prefix + k: v.detach().cpu().float().clone() for k, v in mod.state_dict().items()}


def lens_mask(lengths, T):
    l = torch.tensor(lengths)
    return (torch.arange(T)[None, :] < l[:, None]).unsqueeze(1).float()


def _check_param_grads(mod, P, prefix, tol=8e-2, skip=()):
    worst = ("", 0.0)
    n = 0
    for name, p in mod.named_parameters():
        ref = P[prefix + name].grad
        if ref is None or any(name.startswith(s) for s in skip):
            continue
        assert p.grad is not None, name
        if ref.abs().max().item() < 1e-7:
            assert p.grad.abs().max().item() < 1e-4, name
            continue
        e = relerr(p.grad.cpu(), ref)
        n += 1
        if e > worst[1]:
            worst = (name, e)
        assert e < tol, (name, e)
    assert n > 0
    return worst


def test_dds_conv_module_fwd_bwd(built):
    """DilatedDepthSeparableConv as a drop-in module (modules.py:718-735) on the golden's inputs: output, input / cond
    gradients, every parameter gradient; dilations 1, 3, 9 cross the utterance borders of the rows layout."""
    from glow_tts_amd import predictors
    dds = fill_module(predictors.DilatedDepthSeparableConv(192, 3, 3, 0.5), "dds.").eval()
    P = {k: v.requires_grad_(True) for k, v in cpu_state(dds, "dds.").items()}
    x, g, m = t("dds_x").clone().requires_grad_(True), t("dds_g").clone().requires_grad_(True), t("f1_mask")
    o = R.dds_conv(P, "dds.", x, m, g=g)
    r = torch.randn(o.shape, generator=torch.Generator().manual_seed(3)) * m
    (o * r).sum().backward()
    dds = dds.to(dev())
    xd, gd = x.detach().to(dev()).requires_grad_(True), g.detach().to(dev()).requires_grad_(True)
    od = dds(xd, m.to(dev()), g=gd)
    assert relerr(od.detach().cpu(), t("dds_out")) < 2e-2 and relerr(od.detach().cpu(), o.detach()) < 2e-2
    (od * r.to(dev())).sum().backward()
    vm = m.bool().expand_as(x)
    assert relerr(xd.grad.cpu()[vm], x.grad[vm]) < 3e-2 and relerr(gd.grad.cpu()[vm], g.grad[vm]) < 3e-2
    _check_param_grads(dds, P, "dds.")


def test_dds_dropout_replays_in_backward(built):
    """The dropout of modules.py:733 inside gt_dds_out_fwd is a counter-based mask of (seed, row, channel): the mask read
    off the forward (train output vs evaluation output) is exactly the one gt_dds_out_bwd applies — d h2 equals autograd
    through x + mask / (1 - p) * gelu(LayerNorm(h2)) with that mask (fp32 kernels: 1e-4)."""
    import torch.nn.functional as F
    from glow_tts_amd import _lib, ops
    L = _lib.lib()
    R_, C, p, seed = 70, 192, 0.5, 1234
    g = torch.Generator().manual_seed(2)
    h2 = torch.randn(R_, C, generator=g).to(dev()); x = torch.randn(R_, C, generator=g).to(dev()); dy = torch.randn(R_, C, generator=g).to(dev())
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev()); beta = (0.1 * torch.randn(C, generator=g)).to(dev())
    mask = torch.ones(R_, device=dev()); mask[:3] = 0; mask[40:44] = 0
    st = _lib.current_stream(dev())
    outs = []
    for pp in (p, 0.0):
        o = torch.empty(R_, C, device=dev())
        _lib.check(L.gt_dds_out_fwd(_lib.ptr(h2), _lib.ptr(x), C, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mask), _lib.ptr(o), None, R_, C, 1e-5,
                                    pp, seed, None, st), "gt_dds_out_fwd")
        outs.append(o)
    y_eval = outs[1] - x * mask[:, None]
    keep = ((outs[0] - x * mask[:, None]).abs() > 0) | (y_eval.abs() < 1e-12)
    frac = keep[mask.bool()].float().mean().item()
    assert 0.42 < frac < 0.58, frac
    assert torch.allclose(outs[0], (x + keep * y_eval * 2.0) * mask[:, None], atol=1e-5)
    hh = h2.clone().requires_grad_(True)
    ref = (x + keep * 2.0 * F.gelu(F.layer_norm(hh, (C,), gamma, beta, 1e-5))) * mask[:, None]
    (ref * dy).sum().backward()
    dh = torch.empty(R_, 3 * C, dtype=torch.bfloat16, device=dev())
    dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    _lib.check(L.gt_dds_out_bwd(_lib.ptr(h2), _lib.ptr(dy), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mask), _lib.ptr(dh), _lib.ptr(dg), _lib.ptr(db),
                                None, R_, C, 1e-5, p, seed, None, st), "gt_dds_out_bwd")
    got = dh[:, :C].float() + dh[:, 2 * C:].float()                   # bf16x3 layout: hi + lo
    assert torch.equal(dh[:, :C], dh[:, C:2 * C])
    assert relerr(got, hh.grad) < 1e-4, relerr(got, hh.grad)
    # the partials form (per-workgroup rows, summed by gt_param_partials_reduce) gives the same parameter gradients as the atomics
    import ctypes
    part = torch.empty(L.gt_dds_bwd_partial_rows(R_), 2 * C, device=dev())
    _lib.check(L.gt_dds_out_bwd(_lib.ptr(h2), _lib.ptr(dy), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mask), _lib.ptr(dh), None, None,
                                _lib.ptr(part), R_, C, 1e-5, p, seed, None, st), "gt_dds_out_bwd")
    dg2, db2 = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
    args = _lib.PartialsArgs()
    j = args.job[0]
    j.partials, j.dst_a, j.dst_b, j.n_rows, j.Ca, j.Cb = part.data_ptr(), dg2.data_ptr(), db2.data_ptr(), part.shape[0], C, C
    args.n_jobs = 1
    _lib.check(L.gt_param_partials_reduce(ctypes.byref(args), st), "gt_param_partials_reduce")
    assert relerr(dg2, dg) < 1e-5 and relerr(db2, db) < 1e-5


@pytest.mark.parametrize("which", ["sdp", "spp", "sep"])
def test_stochastic_predictor_nll_and_grads(built, which):
    """models.Stochastic{Duration,Pitch,Energy}Predictor.forward (models.py:261-322, 364-396, 438-470) with the golden's
    injected noise: nll per utterance (vs the REFERENCE's value in the golden and vs the oracle) and every parameter gradient."""
    from glow_tts_amd import predictors
    if which == "sdp":
        mod = fill_module(predictors.StochasticDurationPredictor(192, 192, 3, 0.5, 4, gin_channels=512, lin_channels=4), "sdp.").eval()
        x, m, dr, nz = t("p5_x"), t("f1_mask"), t("p5_w"), t("p5_ew")
        kw = dict(g=t("p5_g"), l=t("p5_l"))
    else:
        cls = predictors.StochasticPitchPredictor if which == "spp" else predictors.StochasticEnergyPredictor
        mod = fill_module(cls(192, 256, 3, 0.1, 4, gin_channels=512), which + ".").eval()
        x, m, nz = t("p5_xf"), t("p5_fmask"), t("p5_ep")
        dr = t("p5_pitch") if which == "spp" else t("p5_pitch").abs()
        kw = dict(g=t("p5_g"))
    P = {k: v.requires_grad_(True) for k, v in cpu_state(mod, which + ".").items()}
    if which == "sdp":
        nll = R.sdp_fwd(P, "sdp.", x, m, dr, nz, **kw)
    else:
        nll = R.spp_fwd(P, which + ".", x, m, dr, nz, **kw)
    w = torch.tensor([1.0, -0.7])
    (nll * w).sum().backward()
    mod = mod.to(dev())
    out = mod(x.to(dev()), m.to(dev()), dr.to(dev()), noise=nz.to(dev()), **{k: v.to(dev()) for k, v in kw.items()})
    want = t(which + "_nll")
    assert relerr(out.detach().cpu(), want) < 1e-2, (out, want)
    assert relerr(out.detach().cpu(), nll.detach()) < 1e-2
    (out * w.to(dev())).sum().backward()
    worst = _check_param_grads(mod, P, which + ".")
    print(which, "worst parameter-gradient error", worst)


@pytest.mark.parametrize("which", ["spp", "sep"])
def test_stochastic_predictor_at_frame_rate_bench_size(built, which):
    """The pitch / energy predictors at the bench's size (cfg 5: B = 32, T_y <= 400 frames -> ~9 k frame-rate rows; VERDICT r2: only
    toy sizes were parity-tested): nll of two utterances against the oracle run on just those utterances (rows of different
    utterances never mix: DDSConv's dilated taps stop at the utterance's zero halo, the nll is summed per utterance)."""
    from glow_tts_amd import predictors
    cls = predictors.StochasticPitchPredictor if which == "spp" else predictors.StochasticEnergyPredictor
    mod = fill_module(cls(192, 256, 3, 0.1, 4, gin_channels=512), which + ".").eval()
    P = cpu_state(mod, which + ".")
    B, T = 32, 400
    g = torch.Generator().manual_seed(5)
    lens = (torch.randint(75, 201, (B,), generator=g) * 2).tolist()
    lens[0] = T
    m = lens_mask(lens, T)
    x = torch.randn(B, 192, T, generator=g) * m
    dr = torch.randn(B, 1, T, generator=g) * m
    if which == "sep":
        dr = dr.abs()
    nz = torch.randn(B, 1, T, generator=g) * m
    spk = torch.randn(B, 512, 1, generator=g)
    mod = mod.to(dev())
    out = mod(x.to(dev()), m.to(dev()), dr.to(dev()), noise=nz.to(dev()), g=spk.to(dev()))
    assert out.shape[0] == B and torch.isfinite(out).all()
    for u in (0, 13, 31):
        Tu = lens[u]
        nll = R.spp_fwd(P, which + ".", x[u:u + 1, :, :Tu], m[u:u + 1, :, :Tu], dr[u:u + 1, :, :Tu], nz[u:u + 1, :, :Tu], g=spk[u:u + 1])
        assert abs(out[u].item() - nll.item()) < 1e-2 * abs(nll.item()) + 1e-2, (u, out[u].item(), nll.item())


def test_stochastic_predictors_reverse(built):
    """reverse=True (synthesis) branches vs the reference's outputs in the golden (sdp_rev, spp_rev)."""
    from glow_tts_amd import predictors
    sdp = fill_module(predictors.StochasticDurationPredictor(192, 192, 3, 0.5, 4, gin_channels=512, lin_channels=4), "sdp.").eval().to(dev())
    out = sdp(t("p5_x").to(dev()), t("f1_mask").to(dev()), g=t("p5_g").to(dev()), l=t("p5_l").to(dev()), reverse=True, noise_scale=0.8,
              noise=t("p5_ew").to(dev()))
    m = t("f1_mask").bool()
    assert relerr(out.cpu()[m], t("sdp_rev")[m]) < 3e-2
    spp = fill_module(predictors.StochasticPitchPredictor(192, 256, 3, 0.1, 4, gin_channels=512), "spp.").eval().to(dev())
    nz = torch.cat([t("p5_ep"), t("p5_ep").flip(2)], 1)
    out = spp(t("p5_xf").to(dev()), t("p5_fmask").to(dev()), g=t("p5_g").to(dev()), reverse=True, noise_scale=0.7, noise=nz.to(dev()))
    fm = t("p5_fmask").bool()
    assert relerr(out.cpu()[fm], t("spp_rev")[fm]) < 3e-2


CFG5 = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1, n_blocks_dec=12,
            n_layers_enc=10, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5, n_block_layers=4, n_sqz=2,
            prenet=True, mean_only=True, hidden_channels_enc=192, hidden_channels_dec=192, window_size=4, gin_channels=512,
            use_sdp=True, use_spk_embeds=True, use_lang_embeds=True, use_emo_embeds=True, lin_channels=4, emoin_channels=1024,
            use_spp=True, use_sep=True)      # == configs/base_blank_emo_lang_pitch.json "model"


def _cfg5_inputs(B, Tx, Ty, seed):
    g = torch.Generator().manual_seed(seed)
    xl = torch.randint(max(2, Tx // 2), Tx + 1, (B,), generator=g); xl[0] = Tx
    yl = torch.maximum(torch.randint(Ty // 3, Ty // 2 + 1, (B,), generator=g) * 2, xl + xl % 2); yl[0] = Ty
    ids = torch.randint(1, 187, (B, Tx), generator=g) * (torch.arange(Tx)[None, :] < xl[:, None])
    ym = lens_mask(yl.tolist(), Ty)
    y = torch.randn(B, 80, Ty, generator=g) * ym
    graw = torch.randn(B, 512, generator=g)
    emo = torch.randint(0, 5, (B,), generator=g)
    cart = torch.rand(B, 3, generator=g) * torch.tensor([1.5, 3.1, 4.6]) + torch.tensor([0.0, 0.0, -1.55])     # inside the bin tables (<= pi)
    pitch = ((80 + 200 * torch.rand(B, 1, Ty, generator=g)) * (torch.rand(B, 1, Ty, generator=g) > 0.3)) * ym
    energy = (1 + 10 * torch.rand(B, 1, Ty, generator=g)) * ym
    lid = torch.randint(0, 3, (B,), generator=g)
    noise = (torch.randn(B, 2, Tx, generator=g), torch.randn(B, 1, Ty, generator=g), torch.randn(B, 1, Ty, generator=g))
    return ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, noise


@pytest.mark.parametrize("ragged", [False, True])
def test_cfg5_flow_generator_forward_backward_vs_oracle(built, ragged):
    """configs/base_blank_emo_lang_pitch.json: FlowGenerator(**hps.model) called as train_ms_emo_lang_pitch.py:284-289 calls it
    (g raw 512-d embedding, emo ids, emo_cartesian, pitch, energy, language ids), here with 2 decoder blocks / 2 encoder
    layers for the CPU oracle's sake: the 5-tuple, the training loss, and EVERY parameter gradient of the model."""
    from glow_tts_amd import models
    cfg = dict(CFG5, n_blocks_dec=2, n_layers_enc=3)
    gen = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **cfg), "").eval()
    P = {k: v.requires_grad_(v.dtype.is_floating_point and "bins" not in k) for k, v in cpu_state(gen).items()}
    ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, noise = _cfg5_inputs(3, 21, 46, seed=9)
    gen = gen.to(dev())
    gen.rows_cfg.ragged = ragged
    d = lambda v: v.to(dev())                                         # noqa: E731
    (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, l_length, l_pitch, l_energy), _, _ = \
        gen(d(ids), d(xl), d(y), d(yl), g=d(graw), emo=d(emo), emo_cartesian=d(cart), pitch=d(pitch), energy=d(energy), l=d(lid),
            noise=tuple(d(n) for n in noise))
    l_mle = models.mle_loss(z, z_m, z_logs, logdet, z_mask)
    loss = l_mle + torch.sum(l_length) + 0.5 * l_pitch + 0.5 * l_energy           # train_ms_emo_lang_pitch.py:295-306
    loss.backward()
    rcx, xb = gen.encoder._last_rows                                   # the predictors' (detached) input, as the product stores it
    out = R.train_forward_full(P, ids, xl, y, yl, lambda logp, mk: attn.squeeze(1).cpu().float(), cfg, graw, emo, cart, pitch, energy,
                               lid, noise, x_for_predictors=rcx.from_rows(xb.float()).cpu())
    out["loss"].backward()
    from oracle import mas as omas
    with torch.no_grad():
        ref_path = torch.from_numpy(omas.oracle_maximum_path(out["logp"].numpy(), (out["x_mask"].unsqueeze(-1) * out["z_mask"].unsqueeze(2)).squeeze(1).numpy()))
    assert (ref_path != attn.squeeze(1).cpu()).float().sum().item() <= 0.1 * ref_path.sum().item()      # bf16 logp: a few frames may move
    assert relerr(z.detach().cpu(), out["z"].detach()) < 3e-2
    assert relerr(l_length.detach().cpu(), out["l_length"].detach()) < 2e-2, (l_length, out["l_length"])
    assert abs(l_pitch.item() - out["l_pitch"].item()) < 2e-2 * max(1.0, abs(out["l_pitch"].item())), (l_pitch, out["l_pitch"])
    assert abs(l_energy.item() - out["l_energy"].item()) < 2e-2 * max(1.0, abs(out["l_energy"].item())), (l_energy, out["l_energy"])
    assert abs(loss.item() - out["loss"].item()) < 2e-2 * max(1.0, abs(out["loss"].item())), (loss.item(), out["loss"].item())
    bad = []
    for name, p in gen.named_parameters():
        if not p.requires_grad:
            continue
        ref = P[name].grad
        assert p.grad is not None, name
        if ref is None or ref.abs().max().item() < 1e-7:
            continue
        a, b = p.grad.cpu().double(), ref.double()
        e = (a - b).norm().item() / max(1e-9, b.norm().item())
        if name.endswith("cond_layer1.weight_v"):                   # one input channel: a mathematically zero gradient (rounding noise)
            continue
        # ReLU paths of the text encoder (prenet, FFN) flip units at bf16 rounding: 0.2 there, 0.1 elsewhere
        # emb_l.weight: per-utterance sums of ~1e-3 entries; the oracle itself moves it by 0.47 when its GEMM operands are
        # rounded to bf16 (tools/bf16_noise_oracle.py) — checked for the right order of magnitude only
        if name == "emb_l.weight":
            assert e < 1.0, e
            continue
        if e > (0.2 if (".pre.conv" in name or "emb_rel" in name or name == "encoder.emb.weight") else 0.1):
            bad.append((name, e))
    assert not bad, bad[:10]
    for key in ("emb_g.weight", "emo_id_proj.weight", "emo_proj.weight", "emo_VAD_inten_proj.weight", "elevation_emb.weight",
                "azimuth_emb.weight", "sty_proj.weight", "emosty_layer_norm.weight", "emb_l.weight", "proj_pitch.flows.2.proj.weight",
                "proj_energy.cond.weight", "encoder.proj_w.post_flows.1.convs.convs_sep.2.weight"):
        assert dict(gen.named_parameters())[key].grad.abs().max().item() > 0, key


def test_cfg5_full_model_against_the_reference_golden(built):
    """The FULL cfg-5 model (12 blocks x 3 WaveNets, 10 encoder layers, 86 913 010 parameters) on the golden's inputs against the
    outputs of the reference's own FlowGenerator.forward (full_* arrays).  Two runs:
      (1) free-running: z, log-det and the alignment the model searches on its own lattice (bf16 z may move a frame or two of the
          MAS path relative to the fp32 reference: at most 10 % of the path's frames may differ);
      (2) with the REFERENCE's alignment injected (forward(path=full_attn), the hook beside noise=): everything downstream of the
          path is then comparable whatever (1) found — the three predictor losses, the training loss and every parameter gradient
          the golden holds are checked unconditionally (round 2 checked them only when the searched path happened to
          match, and said nothing when it did not)."""
    from glow_tts_amd import models
    gen = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **CFG5), "").eval().to(dev())
    assert sum(p.numel() for p in gen.parameters()) == 86913010                 # models.FlowGenerator(**hps.model) of the reference
    d = lambda k: t(k).to(dev())                                       # noqa: E731
    inputs = dict(g=d("full_g"), emo=d("full_emo"), emo_cartesian=d("full_cart"), pitch=d("full_pitch"), energy=d("full_energy"), l=d("full_lid"),
                  noise=(d("full_nw"), d("full_np"), d("full_ne")))
    n_el = (t("full_yl") // 2 * 160).float()
    # (1) free-running
    with torch.no_grad():
        (z, z_m, z_logs, logdet, z_mask), _, (attn, *_), _, _ = gen(d("full_ids"), d("full_xl"), d("full_y"), d("full_yl"), **inputs)
    moved = (attn.cpu() != t("full_attn")).float().sum().item() / 2.0          # a frame that moves clears one cell and sets another
    print(f"free-running MAS path: {int(moved)} of {int(t('full_attn').sum().item())} frames differ from the reference's")
    assert moved <= 0.1 * t("full_attn").sum().item()
    assert relerr(z.detach().cpu(), t("full_z")) < 3e-2
    assert ((logdet.detach().cpu() - t("full_logdet")).abs() < 2e-3 * n_el + 1e-2).all()
    # (2) the reference's alignment injected
    (z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, l_pitch, l_energy), _, _ = \
        gen(d("full_ids"), d("full_xl"), d("full_y"), d("full_yl"), path=d("full_attn"), **inputs)
    assert torch.equal(attn.cpu(), t("full_attn"))
    l_mle = models.mle_loss(z, z_m, z_logs, logdet, z_mask)
    loss = l_mle + torch.sum(l_length) + 0.5 * l_pitch + 0.5 * l_energy
    loss.backward()
    assert relerr(z.detach().cpu(), t("full_z")) < 3e-2
    assert relerr(l_length.detach().cpu(), t("full_l_length")) < 2e-2, (l_length, t("full_l_length"))
    assert abs(l_pitch.item() - t("full_l_pitch").item()) < 2e-2 * abs(t("full_l_pitch").item()) + 1e-2
    assert abs(l_energy.item() - t("full_l_energy").item()) < 2e-2 * abs(t("full_l_energy").item()) + 1e-2
    assert abs(loss.item() - t("full_loss").item()) < 2e-2 * abs(t("full_loss").item()) + 1e-2
    params = dict(gen.named_parameters())
    checked = 0
    for k in G.files:
        if not k.startswith("full_g_"):
            continue
        name, ref = k[len("full_g_"):], t(k)
        got = params[name].grad.cpu()
        e = (got.double() - ref.double()).norm().item() / max(1e-9, ref.double().norm().item())
        assert e < 0.12, (name, e)
        checked += 1
    assert checked >= 20, checked


def test_cfg5_trainer_eager_and_graph_steps(built):
    """cfg 5 through the trainer (train_ms_emo_lang_pitch.py:281-314), eager and captured (frame-rate rows context of the
    pitch / energy predictors included; the predictors' noise is drawn inside the step, so the two are not compared
    number by number): finite losses, one Adam step per step(), every predictor / front-end parameter moves."""
    from glow_tts_amd import train
    cfg = dict(CFG5, n_blocks_dec=2, n_layers_enc=2, p_dropout=0.0, p_dropout_dec=0.0, n_lang=10)

    def make():
        torch.manual_seed(0)
        m = train.build_model(cfg, n_vocab=187, device=dev())
        fill_module(m, "")
        m.encoder.pre.p_dropout = 0.0
        m.encoder.proj_w.convs.dropout_p = m.encoder.proj_w.post_convs.dropout_p = 0.0
        m.proj_pitch.convs.dropout_p = m.proj_energy.convs.dropout_p = 0.0
        return m
    m1, m2 = make(), make()
    before = {n: p.detach().clone() for n, p in m1.named_parameters()}
    ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, _ = _cfg5_inputs(4, 30, 80, seed=4)
    d = lambda v: v.to(dev())                                         # noqa: E731
    kw = dict(g=d(graw), emo=d(emo), emo_cartesian=d(cart), pitch=d(pitch), energy=d(energy), l=d(lid))
    t1, t2 = train.Trainer(m1, graph=False), train.Trainer(m2, graph=True)
    torch.manual_seed(5)
    for _ in range(2):
        l1, _ = t1.step(d(ids), d(xl), d(y), d(yl), lengths_host=(xl.tolist(), yl.tolist()), **kw)
    torch.manual_seed(5)
    for _ in range(2):
        l2, _ = t2.step(d(ids), d(xl), d(y), d(yl), lengths_host=(xl.tolist(), yl.tolist()), **kw)
    torch.cuda.synchronize()
    assert t2.graph_mode and t2.n_captures == 1 and t2.adam_steps == 2
    assert torch.isfinite(l1).item() and torch.isfinite(l2).item()
    assert abs(l1.item() - l2.item()) < 0.5 * max(1.0, abs(l1.item()))          # same model, same batch, different noise draws
    for mm in (m1, m2):
        moved = {n for n, p in mm.named_parameters() if (p.detach() - before[n]).abs().max().item() > 0}
        _assert_moved(moved)


def _assert_moved(moved):
    for key in ("emb_g.weight", "emo_proj.bias", "emosty_layer_norm.weight", "encoder.proj_w.flows.3.proj.weight",
                "encoder.proj_w.post_flows.0.log_scale", "proj_pitch.flows.1.convs.norms_1.0.gamma", "proj_energy.pre.weight",
                "decoder.flows.2.wn_pitch.in_layers.0.weight_v"):
        assert key in moved, key


def test_cfg5_infer_runs_the_predictors_in_reverse(built):
    """FlowGenerator.infer for cfg 5 (models.py:1135-1231): speaker / emotion front end, the StochasticDurationPredictor in
    reverse for the durations, predicted pitch / energy contours into the reverse decoder; the 4-tuple of the reference."""
    from glow_tts_amd import models
    cfg = dict(CFG5, n_blocks_dec=2, n_layers_enc=2)
    gen = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **cfg), "").eval().to(dev())
    ids, xl, y, yl, graw, emo, cart, pitch, energy, lid, _ = _cfg5_inputs(2, 15, 40, seed=2)
    d = lambda v: v.to(dev())                                         # noqa: E731
    gen.store_inverse()
    (mel, z_m, z_logs, ld, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_), (pit, ene) = \
        gen.infer(d(ids), d(xl), g=d(graw), emo=d(emo), emo_cartesian=d(cart), l=d(lid), noise_scale=0.5, length_scale=1.0)
    Ty = mel.shape[2]
    assert ld is None and mel.shape[:2] == (2, 80) and pit.shape == (2, Ty) and ene.shape == (2, Ty)
    assert torch.isfinite(mel).all() and torch.isfinite(pit).all() and torch.isfinite(ene).all() and torch.isfinite(logw).all()
    assert torch.equal(attn.squeeze(1).sum(1), z_mask.squeeze(1))            # every valid frame belongs to exactly one token


def test_voice_conversion_round_trip(built):
    """FlowGenerator.voice_conversion (models.py:1233-1247): decoder forward under the source speaker's vector, reverse under the
    target's.  Same speaker on both sides => the mel comes back (5e-3 of max|y|); another target => another mel.  The reference feeds
    emb_g's output straight to the decoder (no emotion half), so the method only fits a model whose emb_g is gin_channels wide — the
    fork's cfg 5 (emb_g: 512 -> gin / 2) raises a shape error there, and so does this class; the test widens emb_g by hand."""
    from glow_tts_amd import models
    cfg = dict(CFG5, n_blocks_dec=3, n_layers_enc=1, gin_channels=64, use_emo_embeds=False, use_spp=False, use_sep=False, use_sdp=False)
    gen = models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **cfg)
    gen.emb_g = torch.nn.Linear(512, 64)
    gen = fill_module(gen, "").eval().to(dev())
    for b in range(3):                                                    # a coupling that does something (end is zero-initialised)
        torch.nn.init.normal_(gen.decoder.flows[3 * b + 2].end.weight, std=0.02)
    gen.store_inverse()
    g = torch.Generator().manual_seed(11)
    yl = torch.tensor([40, 32])
    y = (torch.randn(2, 80, 40, generator=g) * lens_mask(yl.tolist(), 40)).to(dev())
    e_src, e_tgt = torch.randn(2, 512, generator=g).to(dev()), torch.randn(2, 512, generator=g).to(dev())
    same = gen.voice_conversion(y, yl.to(dev()), e_src, e_src)
    assert same.shape == y.shape and (same - y).abs().max().item() < 5e-3 * y.abs().max().item()
    other = gen.voice_conversion(y, yl.to(dev()), e_src, e_tgt)
    assert torch.isfinite(other).all() and (other - y).abs().max().item() > 1e-2
    assert other[1, :, 32:].abs().max().item() == 0.0                     # padded frames stay zero
    with pytest.raises(Exception):                                        # the fork's own cfg 5 shape: emb_g is gin / 2 wide
        bad = fill_module(models.FlowGenerator(n_vocab=187, out_channels=80, n_lang=10, **dict(cfg, gin_channels=64)), "").eval().to(dev())
        bad.voice_conversion(y, yl.to(dev()), e_src, e_tgt)
