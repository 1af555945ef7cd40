"""GPU parity tests: flow-decoder HIP path (through the C-ABI, glow_tts_amd modules) vs the float
oracle (oracle/glowtts_ref.py, CPU fp32, pinned to the reference by tests/golden/float_golden.npz).

Tolerances (stated, north_star "stated fp tolerance"): GEMMs run in bf16 with fp32 accumulation and
bf16 hidden activations, so z / gradients are compared at 3e-2 of the tensor's max-abs (observed
~5e-3); log-dets at 2e-3 * valid frames; the fp32-only kernels (ActNorm/InvConvNear) at 1e-4."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from fill import fill_module  # noqa: E402
from oracle import glowtts_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


def lens_mask(lengths, T):
    l = torch.tensor(lengths)
    return (torch.arange(T)[None, :] < l[:, None]).unsqueeze(1).float()


def cpu_state(mod, prefix=""):
    return {prefix + k: v.detach().cpu().float() for k, v in mod.state_dict().items()}


def test_wgrad_kernel_vs_torch(built):
    from glow_tts_amd import _lib, ops, flow_impl
    from glow_tts_amd.modules import ConvP, WNConvP
    for (Cin, Cout, k, wn) in [(192, 384, 5, True), (192, 160, 1, False), (80, 192, 1, True), (192, 768, 3, False)]:
        B, T = 3, 90
        g = torch.Generator().manual_seed(Cin + k)
        ctx = ops.RowsCtx(torch.tensor([90, 41, 7], dtype=torch.int32, device=dev()), T)
        m = ctx.rowmask2d[:, ops.HALO:ops.HALO + T].unsqueeze(1)
        x = (torch.randn(B, Cin, T, generator=g).to(dev()) * m).to(torch.bfloat16)
        dy = (torch.randn(B, Cout, T, generator=g).to(dev()) * m).to(torch.bfloat16)
        conv = (WNConvP if wn else ConvP)(Cin, Cout, k).to(dev())
        conv.prepare()
        grads = flow_impl.conv_param_grads(conv, ctx.to_rows(x), ctx.to_rows(dy), ctx.R)
        # torch reference on the same bf16-rounded operands
        if wn:
            v = conv.weight_v.detach().clone().requires_grad_(True); gg = conv.weight_g.detach().clone().requires_grad_(True)
            w = gg * v / v.reshape(Cout, -1).norm(dim=1).reshape(Cout, 1, 1)
        else:
            v = conv.weight.detach().clone().requires_grad_(True); w = v
        b = conv.bias.detach().clone().requires_grad_(True)
        F.conv1d(x.float(), w, b, padding=k // 2).backward(dy.float())
        if wn:
            assert relerr(grads[conv.weight_v], v.grad) < 1e-2 and relerr(grads[conv.weight_g], gg.grad) < 1e-2
        else:
            assert relerr(grads[conv.weight], v.grad) < 1e-2
        assert relerr(grads[conv.bias], b.grad) < 1e-3


def test_wgrad_ring_long_rows_and_ragged_ends(built):
    """The weight-gradient kernel's operand ring (LDS-DMA stages of 32 rows, conv_wgrad.hip) on inputs long enough to run its steady state:
    several slabs per job, a row count that is no multiple of the stage (the last stage is the zero-filled register path), channel counts
    that are no multiple of the 128 x 64 tile (clamped chunks), every tap count; immediate mode and the batched queue (two dY pieces)."""
    from glow_tts_amd import ops, flow_impl, wgrad
    from glow_tts_amd.modules import ConvP, WNConvP
    B, T = 5, 777
    lens = [777, 640, 333, 100, 9]
    for (Cin, Cout, k, wn) in [(192, 384, 5, True), (80, 192, 1, True), (192, 160, 1, False), (192, 768, 3, False), (192, 384, 1, True)]:
        g = torch.Generator().manual_seed(Cin + 7 * k)
        ctx = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev()), T)
        m = ctx.rowmask2d[:, ops.HALO:ops.HALO + T].unsqueeze(1)
        x = (torch.randn(B, Cin, T, generator=g).to(dev()) * m).to(torch.bfloat16)
        dy = (torch.randn(B, Cout, T, generator=g).to(dev()) * m).to(torch.bfloat16)
        conv = (WNConvP if wn else ConvP)(Cin, Cout, k).to(dev())
        conv.prepare()
        xr, dyr = ctx.to_rows(x), ctx.to_rows(dy)
        got = [flow_impl.conv_param_grads(conv, xr, dyr, ctx.R)]
        with wgrad.WgradQueue(dev()):                                          # batched, dY in two column pieces
            h = Cout // 2
            got.append(flow_impl.conv_param_grads(conv, xr, None, ctx.R, parts=[(dyr[:, :h], 0, h), (dyr[:, h:], h, Cout - h)]))
        if wn:
            v = conv.weight_v.detach().clone().requires_grad_(True); gg = conv.weight_g.detach().clone().requires_grad_(True)
            w = gg * v / v.reshape(Cout, -1).norm(dim=1).reshape(Cout, 1, 1)
        else:
            v = conv.weight.detach().clone().requires_grad_(True); w = v
        b = conv.bias.detach().clone().requires_grad_(True)
        F.conv1d(x.float(), w, b, padding=k // 2).backward(dy.float())
        for grads in got:
            if wn:
                assert relerr(grads[conv.weight_v], v.grad) < 1e-2 and relerr(grads[conv.weight_g], gg.grad) < 1e-2, (Cin, Cout, k)
            else:
                assert relerr(grads[conv.weight], v.grad) < 1e-2, (Cin, Cout, k)
            assert relerr(grads[conv.bias], b.grad) < 1e-3, (Cin, Cout, k)


def test_actnorm_invconv_fwd_bwd(built):
    from glow_tts_amd import ops, flow_impl
    B, T, C = 2, 40, 160
    g = torch.Generator().manual_seed(5)
    lens = [40, 23]
    rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev()), T)
    m = lens_mask(lens, T)
    x = torch.randn(B, C, T, generator=g) * m
    logs = (torch.randn(1, C, 1, generator=g) * 0.2).requires_grad_(True)
    bias = (torch.randn(1, C, 1, generator=g) * 0.2).requires_grad_(True)
    W = (torch.eye(4) + 0.3 * torch.randn(4, 4, generator=g)).requires_grad_(True)
    xx = x.clone().requires_grad_(True)
    P = {"a.logs": logs, "a.bias": bias, "i.weight": W}
    z1, ld1 = R.actnorm_fwd(P, "a.", xx, m)
    z, ld2 = R.invconv_fwd(P, "i.", z1, m)
    rz = torch.randn(z.shape, generator=g); rl = torch.randn(B, generator=g)
    ((z * rz).sum() + ((ld1 + ld2) * rl).sum()).backward()
    # HIP
    d = dev()
    logdet = torch.zeros(B, device=d)
    xr = rc.to_rows(x.to(d))
    y, x0, saved = flow_impl.actnorm_invconv_fwd(rc, xr, logs.detach().to(d), bias.detach().to(d), W.detach().to(d), logdet)
    assert relerr(rc.from_rows(y).cpu(), z.detach()) < 1e-5
    assert relerr(logdet.cpu(), (ld1 + ld2).detach()) < 1e-5
    assert relerr(rc.from_rows(x0.float()).cpu(), z.detach()[:, :C // 2]) < 1e-2
    lg, bs, Wd = logs.detach().to(d), bias.detach().to(d), W.detach().to(d)
    dx, grads = flow_impl.actnorm_invconv_bwd(rc, saved, rc.to_rows(rz.to(d)), rl.to(d), lg, bs, Wd)
    assert relerr(rc.from_rows(dx).cpu(), xx.grad) < 1e-4
    assert relerr(grads[lg].cpu(), logs.grad) < 1e-4
    assert relerr(grads[bs].cpu(), bias.grad) < 1e-4
    assert relerr(grads[Wd].cpu(), W.grad) < 1e-4


def _run_decoder_case(n_blocks, B, T, lens, seed, check_params, sigmoid_scale=False):
    from glow_tts_amd import models
    torch.manual_seed(seed)
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, n_blocks, 4, p_dropout=0.05, sigmoid_scale=sigmoid_scale), "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    for v in P.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(seed)
    m = lens_mask(lens, T)
    y = torch.randn(B, 80, T, generator=g) * m
    yy = y.clone().requires_grad_(True)
    z, ld = R.decoder_fwd(P, "decoder.", yy, m, n_blocks=n_blocks, sigmoid_scale=sigmoid_scale)
    rz = torch.randn(z.shape, generator=g) * m[:, :, :z.shape[2]]; rl = torch.randn(B, generator=g) * 0.1
    ((z * rz).sum() + (ld * rl).sum()).backward()

    dec = dec.to(dev())
    yd = y.to(dev()).requires_grad_(True)
    zd, ldd = dec(yd, m.to(dev()))
    assert zd.shape == z.shape and ldd.shape == ld.shape
    assert relerr(zd.detach().cpu(), z.detach()) < 3e-2, relerr(zd.detach().cpu(), z.detach())
    nvalid = torch.tensor(lens, dtype=torch.float32) // 2 * 160
    assert ((ldd.detach().cpu() - ld.detach()).abs() < 2e-3 * nvalid + 1e-2).all(), (ldd.cpu(), ld)
    ((zd * rz.to(dev())).sum() + (ldd * rl.to(dev())).sum()).backward()
    assert relerr(yd.grad.cpu(), yy.grad) < 3e-2, relerr(yd.grad.cpu(), yy.grad)
    if check_params:
        worst = {}
        for name, p in dec.named_parameters():
            ref = P["decoder." + name].grad
            assert p.grad is not None, name
            e = relerr(p.grad.cpu(), ref)
            worst[name] = e
            assert e < 6e-2, (name, e)
    return True


def test_decoder_two_blocks_fwd_bwd(built):
    _run_decoder_case(2, 2, 48, [48, 26], seed=3, check_params=True)


def test_decoder_sigmoid_scale(built):
    """sigmoid_scale=True (attentions.py:172-173; oracle pinned by cbs_* in float_golden.npz): forward, log-det, gradients."""
    _run_decoder_case(2, 2, 48, [48, 26], seed=5, check_params=True, sigmoid_scale=True)


def test_decoder_odd_length_and_ragged(built):
    _run_decoder_case(1, 3, 51, [51, 20, 2], seed=4, check_params=False)


def test_decoder_full_depth(built):
    """12 blocks (configs/base.json n_blocks_dec) on a short batch: error does not blow up with depth."""
    _run_decoder_case(12, 2, 64, [64, 30], seed=5, check_params=False)


def test_coupling_block_dropin(built):
    from glow_tts_amd import attentions
    cb = fill_module(attentions.CouplingBlock(160, 192, 5, 1, 4, p_dropout=0.05), "cb.").eval()
    P = cpu_state(cb, "cb.")
    B, T = 2, 30
    m = lens_mask([30, 11], T)
    x = torch.randn(B, 160, T, generator=torch.Generator().manual_seed(8)) * m
    z, ld = R.coupling_fwd(P, "cb.", x, m)
    zd, ldd = cb.to(dev())(x.to(dev()), m.to(dev()))
    assert relerr(zd.cpu(), z) < 3e-2 and (ldd.cpu() - ld).abs().max() < 0.5


def test_decoder_train_mode_dropout_runs(built):
    from glow_tts_amd import models
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05), "decoder.").to(dev()).train()
    m = lens_mask([40, 40], 40).to(dev())
    y = (torch.randn(2, 80, 40, device=dev()) * m).requires_grad_(True)
    z1, ld1 = dec(y, m)
    z2, ld2 = dec(y, m)
    assert torch.isfinite(z1).all() and not torch.equal(z1, z2)        # different dropout masks per call
    (z1.sum() + ld1.sum()).backward()
    assert torch.isfinite(y.grad).all()


@pytest.mark.parametrize("n_blocks,T,lens", [(2, 48, [48, 26]), (12, 65, [65, 30, 2])])
def test_decoder_reverse_vs_oracle_and_round_trip(built, n_blocks, T, lens):
    """Inference direction (models.py:765-785, reverse=True): against the oracle restatement (itself pinned to the
    imported reference by tests/test_float_oracle.py), and the size-independent property reverse(forward(x)) == x:
    both directions evaluate the SAME m / logs from the untouched half, so the round trip only carries fp32 rounding."""
    from glow_tts_amd import models
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, n_blocks, 4, p_dropout=0.05), "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    B = len(lens)
    m = lens_mask(lens, T)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, 80, T, generator=g) * m
    want = R.decoder_rev(P, "decoder.", z, m, n_blocks=n_blocks)
    dec = dec.to(dev())
    x, ld = dec(z.to(dev()), m.to(dev()), reverse=True)
    assert ld is None and x.shape == want.shape
    assert relerr(x.cpu(), want) < 3e-2
    with torch.no_grad():
        z2, _ = dec(x, m.to(dev())[:, :, :x.shape[2]])
    T2 = x.shape[2]
    valid = (m[:, :, :T2] * (torch.arange(T2)[None, None, :] < (torch.tensor(lens) // 2 * 2)[:, None, None])).bool().expand(B, 80, T2)
    err = (z2.cpu() - z[:, :, :T2])[valid].abs().max().item()
    # not bit-exact: an fp32 rounding difference in the untouched half can flip ITS bf16 rounding in front of the WN
    # GEMMs of the next block (measured: 2e-3 after 2 blocks, 9e-3 after 12, on |z| <= 3.8)
    assert err < 5e-3 * max(1.0, z.abs().max().item()), err


def test_infer_generates_mel_through_the_reverse_flow(built):
    """FlowGenerator.infer: predicted durations -> generate_path (== the oracle's, commons.py:127-143) -> prior expansion
    -> reverse decoder; with noise_scale = 0 the mel equals the oracle's reverse decoder of the expanded means."""
    from glow_tts_amd import models
    gen = fill_module(models.FlowGenerator(148, 192, 768, 256, 80, use_sdp=False, kernel_size=3, n_heads=2, n_layers_enc=2, p_dropout=0.1,
                                           n_blocks_dec=2, kernel_size_dec=5, dilation_rate=1, n_block_layers=4,
                                           p_dropout_dec=0.05, n_sqz=2, window_size=4, mean_only=True, prenet=True), "").eval()
    P = cpu_state(gen)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 148, (2, 19), generator=g); xl = torch.tensor([19, 11])
    ids = ids * (torch.arange(19)[None, :] < xl[:, None])
    gen = gen.to(dev())
    (y, z_m, z_logs, ld, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_), (pit, ene) = gen.infer(ids.to(dev()), xl.to(dev()), noise_scale=0.0)
    assert pit is None and ene is None                               # models.py:1231: the 4-tuple of the reference's infer
    assert ld is None and torch.isfinite(y).all() and z_logs.abs().max().item() == 0
    dur = torch.ceil(torch.exp(logw) * x_mask).squeeze(1).cpu()
    assert torch.equal(attn.squeeze(1).sum(-1).cpu(), dur)
    want_attn = R.generate_path(dur, (x_mask.transpose(1, 2) * z_mask).cpu())
    assert torch.equal(attn.squeeze(1).cpu(), want_attn)
    zm_want = torch.matmul(want_attn.transpose(1, 2), x_m.cpu().float().transpose(1, 2)).transpose(1, 2)
    assert torch.allclose(z_m.cpu(), zm_want, atol=1e-5)
    y_want = R.decoder_rev(P, "decoder.", zm_want * z_mask.cpu(), z_mask.cpu(), n_blocks=2)
    assert relerr(y.cpu(), y_want) < 3e-2


@pytest.mark.parametrize("ragged,pitch_only", [(False, False), (True, False), (False, True)])
def test_decoder_pitch_energy_speaker_conditioning(built, ragged, pitch_only):
    """cfg 5's decoder: every coupling block runs wn(g) -> wn_energy(energy) -> wn_pitch(pitch) (attentions.py:152-154),
    the latter two modules.WNP with per-frame conditioning (the gate kernel's per-row cond).  Forward, log-det, input
    gradient, EVERY parameter gradient (cond_layer1 affine maps included), the gradient of g, and the reverse direction
    against the oracle (pinned to the imported reference: decp_* in float_golden.npz); pitch alone leaves wn_energy the
    identity."""
    from glow_tts_amd import models, ops
    n_blocks, B, T, lens = 2, 3, 50, [50, 27, 12]
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, n_blocks, 4, p_dropout=0.05, gin_channels=256, with_prosody_wn=True),
                      "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    for v in P.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(21)
    m = lens_mask(lens, T)
    y = torch.randn(B, 80, T, generator=g) * m
    spk = torch.randn(B, 256, 1, generator=g)
    pit = torch.randn(B, 1, T, generator=g) * m
    ene = None if pitch_only else torch.randn(B, 1, T, generator=g).abs() * m
    yy, gg = y.clone().requires_grad_(True), spk.clone().requires_grad_(True)
    z, ld = R.decoder_fwd(P, "decoder.", yy, m, gg, n_blocks=n_blocks, pitch=pit, energy=ene)
    z0, _ = R.decoder_fwd(P, "decoder.", y, m, spk, n_blocks=n_blocks)
    assert relerr(z0.detach(), z.detach()) > 1e-2                     # the contours matter at these weights
    rz = torch.randn(z.shape, generator=g) * m[:, :, :z.shape[2]]; rl = torch.randn(B, generator=g) * 0.1
    ((z * rz).sum() + (ld * rl).sum()).backward()

    dec = dec.to(dev())
    yd, gd = y.to(dev()).requires_grad_(True), spk.to(dev()).requires_grad_(True)
    dec.rows_cfg = ops.RowsConfig(ragged=ragged)      # a stand-alone decoder: its own rows-layout state
    if ragged:
        dec.rows_cfg.host_lengths["y"] = list(lens)
    try:
        zd, ldd = dec(yd, m.to(dev()), g=gd, pitch=pit.to(dev()), energy=None if ene is None else ene.to(dev()))
        assert relerr(zd.detach().cpu(), z.detach()) < 3e-2, relerr(zd.detach().cpu(), z.detach())
        nvalid = torch.tensor(lens, dtype=torch.float32) // 2 * 160
        assert ((ldd.detach().cpu() - ld.detach()).abs() < 2e-3 * nvalid + 1e-2).all(), (ldd.cpu(), ld)
        ((zd * rz.to(dev())).sum() + (ldd * rl.to(dev())).sum()).backward()
        zrev = torch.randn(B, 80, T, generator=g) * m
        xr, _ = dec(zrev.to(dev()), m.to(dev()), g=spk.to(dev()), pitch=pit.to(dev()),
                    energy=None if ene is None else ene.to(dev()), reverse=True)
    finally:
        dec.rows_cfg = ops.RowsConfig()
    assert relerr(yd.grad.cpu(), yy.grad) < 3e-2, relerr(yd.grad.cpu(), yy.grad)
    assert relerr(gd.grad.cpu(), gg.grad) < 6e-2, relerr(gd.grad.cpu(), gg.grad)
    for name, p in dec.named_parameters():
        ref = P["decoder." + name].grad
        if pitch_only and ".wn_energy." in name:
            assert p.grad is None or p.grad.abs().max().item() == 0, name
            continue
        assert p.grad is not None, name
        if name.endswith("cond_layer1.weight_v"):
            # one input channel: w = g * v / |v| depends on v's sign only — a mathematically zero gradient, rounding noise on both sides
            scale = P["decoder." + name[:-1] + "g"].grad.abs().max().item()
            assert p.grad.abs().max().item() <= 1e-3 * scale and ref.abs().max().item() <= 1e-3 * scale, name
            continue
        e = relerr(p.grad.cpu(), ref)
        assert e < 6e-2, (name, e)
    want = R.decoder_rev(P, "decoder.", zrev, m, spk, n_blocks=n_blocks, pitch=pit, energy=ene)
    assert relerr(xr.cpu(), want.detach()) < 3e-2


def test_coupling_block_standalone_reverse_and_prosody(built):
    """attentions.CouplingBlock as a stand-alone module (attentions.py:132-186): forward with g + pitch + energy incl. the
    gradients of cond_layer1, reverse=True (attentions.py:178-180) and reverse(forward(x)) == x, vs the oracle."""
    from glow_tts_amd import attentions
    cb = fill_module(attentions.CouplingBlock(160, 192, 5, 1, 4, gin_channels=256, p_dropout=0.05, with_prosody_wn=True), "cb.").eval()
    P = {k: v.clone().requires_grad_(True) for k, v in cpu_state(cb, "cb.").items()}
    B, T = 2, 30
    g = torch.Generator().manual_seed(18)
    m = lens_mask([30, 11], T)
    x = torch.randn(B, 160, T, generator=g) * m
    spk = torch.randn(B, 256, 1, generator=g)
    m2 = lens_mask([60, 22], 2 * T)
    pit, ene = torch.randn(B, 1, 2 * T, generator=g) * m2, torch.randn(B, 1, 2 * T, generator=g).abs() * m2
    xx = x.clone().requires_grad_(True)
    z, ld = R.coupling_fwd(P, "cb.", xx, m, spk, pitch=pit, energy=ene)
    rz = torch.randn(z.shape, generator=g) * m
    ((z * rz).sum() + ld.sum() * 0.1).backward()
    cb = cb.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    zd, ldd = cb(xd, m.to(dev()), g=spk.to(dev()), pitch=pit.to(dev()), energy=ene.to(dev()))
    assert relerr(zd.detach().cpu(), z.detach()) < 3e-2 and (ldd.detach().cpu() - ld.detach()).abs().max() < 0.5
    ((zd * rz.to(dev())).sum() + ldd.sum() * 0.1).backward()
    assert relerr(xd.grad.cpu(), xx.grad) < 3e-2
    for name in ("wn_pitch.cond_layer1.weight_g", "wn_pitch.cond_layer1.bias", "wn_energy.cond_layer1.weight_g",
                 "wn_energy.cond_layer1.bias", "wn_energy.in_layers.1.weight_v", "wn.cond_layer.bias"):
        got, want = dict(cb.named_parameters())[name].grad, P["cb." + name].grad
        assert got is not None and relerr(got.cpu(), want) < 6e-2, (name, relerr(got.cpu(), want))
    with torch.no_grad():
        xr, none = cb(zd.detach(), m.to(dev()), reverse=True, g=spk.to(dev()), pitch=pit.to(dev()), energy=ene.to(dev()))
        want = R.coupling_rev({k: v.detach() for k, v in P.items()}, "cb.", z.detach(), m, spk, pitch=pit, energy=ene)
    assert none is None
    assert relerr(xr.cpu(), want) < 3e-2
    assert relerr(xr.cpu(), x) < 3e-2                                   # reverse(forward(x)) == x
    cb.store_inverse()                                                  # frozen images: same answer, no re-pack
    with torch.no_grad():
        xr2, _ = cb(zd.detach(), m.to(dev()), reverse=True, g=spk.to(dev()), pitch=pit.to(dev()), energy=ene.to(dev()))
    assert torch.equal(xr, xr2)


def test_actnorm_ddi_initialises_block_after_block(built):
    """ActNorm data-dependent init (modules.py:588-590, 607-619; configs/base.json "ddi": true): set_ddi(True) on every
    ActNorm, one forward — each layer takes logs / bias from the masked statistics of the input IT sees; later forwards
    leave them alone.  vs the oracle's restatement (pinned by ddi_* in float_golden.npz)."""
    from glow_tts_amd import models, modules
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 3, 4, p_dropout=0.05), "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    B, T, lens = 3, 44, [44, 30, 6]
    m = lens_mask(lens, T)
    y = torch.randn(B, 80, T, generator=torch.Generator().manual_seed(4)) * 1.7 + 0.3
    y = y * m
    P2, z, ld = R.decoder_ddi(P, "decoder.", y, m, n_blocks=3)
    dec = dec.to(dev())
    for f in dec.flows:
        if isinstance(f, modules.ActNorm):
            f.set_ddi(True)
    with torch.no_grad():
        zd, ldd = dec(y.to(dev()), m.to(dev()))
    for b in range(3):
        an = dec.flows[3 * b]
        assert an.initialized
        assert relerr(an.logs.detach().cpu(), P2[f"decoder.flows.{3 * b}.logs"]) < 2e-2, b      # bf16 GEMMs upstream of blocks > 0
        assert relerr(an.bias.detach().cpu(), P2[f"decoder.flows.{3 * b}.bias"]) < 2e-2, b
    assert relerr(dec.flows[0].logs.detach().cpu(), P2["decoder.flows.0.logs"]) < 1e-5           # block 0: fp32 only
    assert relerr(zd.cpu(), z) < 3e-2
    logs0 = dec.flows[3].logs.detach().clone()
    with torch.no_grad():
        dec(y.to(dev()) * 0.5, m.to(dev()))
    assert torch.equal(dec.flows[3].logs.detach(), logs0)


def test_store_inverse_caches_the_synthesis_state(built):
    """FlowSpecDecoder.store_inverse (models.py:787-789): reverse calls after it reuse the packed weights and flow scalars
    and give the same mel; a training-mode forward drops the cache."""
    from glow_tts_amd import models, modules as gm
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 2, 4, p_dropout=0.05), "decoder.").eval().to(dev())
    m = lens_mask([40, 22], 40).to(dev())
    z = torch.randn(2, 80, 40, device=dev()) * m
    x1, _ = dec(z, m, reverse=True)
    dec.store_inverse()
    assert dec._inv_cache is not None and len(dec._inv_cache) == 2
    calls = []
    orig = gm._PackPlan.run
    gm._PackPlan.run = lambda self: (calls.append(1), orig(self))[1]
    try:
        x2, _ = dec(z, m, reverse=True)
    finally:
        gm._PackPlan.run = orig
    assert not calls and torch.equal(x1, x2)
    dec(z, m)
    assert dec._inv_cache is None


def _full_size_case(B, T_y, seed, n_check=3):
    """The 12-block decoder forward + backward on a full BASELINE batch (ragged rows, R ~ 9-10 k), compared for a few
    utterances against the oracle run on just those utterances (utterances are independent through the decoder)."""
    from glow_tts_amd import models, ops
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 12, 4, p_dropout=0.05), "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    g = torch.Generator().manual_seed(seed)
    lens = (torch.randint(T_y * 3 // 16, T_y // 2 + 1, (B,), generator=g) * 2).tolist()
    lens[0] = T_y
    m = lens_mask(lens, T_y)
    y = torch.randn(B, 80, T_y, generator=g) * m
    rz = torch.randn(B, 80, T_y, generator=g) * m
    rl = torch.randn(B, generator=g) * 0.1
    dec = dec.to(dev())
    dec.rows_cfg = ops.RowsConfig(ragged=True, row_round=512)
    dec.rows_cfg.host_lengths["y"] = list(lens)
    yd = y.to(dev()).requires_grad_(True)
    zd, ldd = dec(yd, m.to(dev()))
    ((zd * rz.to(dev())).sum() + (ldd * rl.to(dev())).sum()).backward()
    torch.cuda.synchronize()
    R_rows = sum(v // 2 + 4 for v in lens)
    for u in [0, B // 2, B - 1][:n_check]:
        T = lens[u]
        yy = y[u:u + 1, :, :T].clone().requires_grad_(True)
        z, ld = R.decoder_fwd(P, "decoder.", yy, m[u:u + 1, :, :T], n_blocks=12)
        ((z * rz[u:u + 1, :, :T]).sum() + (ld * rl[u:u + 1]).sum()).backward()
        assert relerr(zd[u:u + 1, :, :T].detach().cpu(), z.detach()) < 3e-2, u
        assert abs(ldd[u].item() - ld.item()) < 2e-3 * (T // 2) * 160 + 1e-2, (u, ldd[u].item(), ld.item())
        assert relerr(yd.grad[u:u + 1, :, :T].cpu(), yy.grad) < 4e-2, (u, relerr(yd.grad[u:u + 1, :, :T].cpu(), yy.grad))
        assert zd[u, :, T:].abs().max().item() == 0 if T < T_y else True
    return R_rows


def test_decoder_full_size_cfg2_batch(built):
    """cfg 2 (configs/base.json): B = 32, T_y <= 800, 12 blocks, ragged R ~ 9 k rows — the shape the bench runs."""
    rows = _full_size_case(32, 800, seed=1234)
    assert rows > 8000


def _full_size_conditioned_case(B, T_y, seed, gin, prosody, check):
    """As _full_size_case with the speaker vector (cfg 4) and, for cfg 5, the three WaveNets per block with per-frame pitch / energy
    conditioning: forward, log-det, input / speaker / contour-independent parameter gradients of a few utterances against the
    oracle run on just those utterances (the per-utterance loss weights make one utterance's parameter-gradient contribution
    separable: all other utterances get weight 0 in a second device pass)."""
    from glow_tts_amd import models, ops
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 12, 4, p_dropout=0.05, gin_channels=gin, with_prosody_wn=prosody), "decoder.").eval()
    P = cpu_state(dec, "decoder.")
    g = torch.Generator().manual_seed(seed)
    lens = (torch.randint(T_y * 3 // 16, T_y // 2 + 1, (B,), generator=g) * 2).tolist()
    lens[0] = T_y
    m = lens_mask(lens, T_y)
    y = torch.randn(B, 80, T_y, generator=g) * m
    spk = torch.randn(B, gin, 1, generator=g)
    pit = torch.randn(B, 1, T_y, generator=g) * m if prosody else None
    ene = torch.randn(B, 1, T_y, generator=g).abs() * m if prosody else None
    rz = torch.randn(B, 80, T_y, generator=g) * m
    rl = torch.randn(B, generator=g) * 0.1
    dec = dec.to(dev())
    dec.rows_cfg = ops.RowsConfig(ragged=True, row_round=512)
    dec.rows_cfg.host_lengths["y"] = list(lens)
    dkw = dict(pitch=pit.to(dev()), energy=ene.to(dev())) if prosody else {}
    yd, gd = y.to(dev()).requires_grad_(True), spk.to(dev()).requires_grad_(True)
    zd, ldd = dec(yd, m.to(dev()), g=gd, **dkw)
    ((zd * rz.to(dev())).sum() + (ldd * rl.to(dev())).sum()).backward()
    torch.cuda.synchronize()
    for u in check:
        T = lens[u]
        yy, gg = y[u:u + 1, :, :T].clone().requires_grad_(True), spk[u:u + 1].clone().requires_grad_(True)
        okw = dict(pitch=pit[u:u + 1, :, :T], energy=ene[u:u + 1, :, :T]) if prosody else {}
        z, ld = R.decoder_fwd(P, "decoder.", yy, m[u:u + 1, :, :T], gg, n_blocks=12, **okw)
        ((z * rz[u:u + 1, :, :T]).sum() + (ld * rl[u:u + 1]).sum()).backward()
        assert relerr(zd[u:u + 1, :, :T].detach().cpu(), z.detach()) < 3e-2, u
        assert abs(ldd[u].item() - ld.item()) < 2e-3 * (T // 2) * 160 + 1e-2, (u, ldd[u].item(), ld.item())
        assert relerr(yd.grad[u:u + 1, :, :T].cpu(), yy.grad) < 4e-2, (u, relerr(yd.grad[u:u + 1, :, :T].cpu(), yy.grad))
        assert relerr(gd.grad[u:u + 1].cpu(), gg.grad) < 6e-2, (u, relerr(gd.grad[u:u + 1].cpu(), gg.grad))
    return sum(v // 2 + 4 for v in lens)


def test_decoder_full_size_cfg4_batch(built):
    """cfg 4 (configs/base_blank_ms.json shapes): B = 20, T_y <= 500, speaker vector gin = 256 — parity at the bench's size
    (cfg 4 / cfg 5 were parity-tested at toy size only)."""
    rows = _full_size_conditioned_case(20, 500, seed=41, gin=256, prosody=False, check=(0, 19))
    assert rows > 2500


def test_decoder_full_size_cfg5_batch(built):
    """cfg 5 (configs/base_blank_emo_lang_pitch.json shapes): B = 32, T_y <= 400, gin = 512, three WaveNets per block with per-frame
    pitch / energy conditioning (the stack kernel's per-row cond at R ~ 5 k rows)."""
    rows = _full_size_conditioned_case(32, 400, seed=43, gin=512, prosody=True, check=(0, 17))
    assert rows > 4000


def test_decoder_full_size_cfg3_batch(built):
    """cfg 3 (configs/base_blank.json shapes): B = 32, T_y <= 872."""
    rows = _full_size_case(32, 872, seed=77, n_check=2)
    assert rows > 8000


@pytest.mark.parametrize("mode", ["plain", "speaker", "per_row"])
def test_fused_wn_layer_kernels_match_the_two_kernel_path(built, mode):
    """gt_wn_layer_fwd / gt_wn_layer_bwd (one kernel per WaveNet layer: k=5 conv + gate + residual 1x1, and the mirror in
    the backward) against round 1's launch sequence (gate conv kernel + 1x1 GEMM kernel per layer) on the same weights,
    inputs and dropout seeds (train mode, p = 0.05: the masks are the same counter hash): outputs, saved T / S, input
    gradient, conditioning gradient and every parameter gradient, ragged rows with R not a multiple of 64."""
    from glow_tts_amd import flow_impl, modules, ops, wgrad
    H, n = 192, 4
    gin = 256 if mode == "speaker" else 0
    wn = fill_module(modules.WN(160, H, 5, 1, n, gin, 0.05), "wn.").to(dev())
    modules.prepare_all(wn)
    lens = [70, 33, 1, 64]
    lt = torch.tensor(lens, dtype=torch.int32, device=dev())
    rc = ops.RowsCtx(lt, 70, lengths_host=lens, round_to=8)
    assert rc.R % 64 != 0
    g = torch.Generator().manual_seed(12)
    h0 = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    dskip = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    cond = None
    if mode == "speaker":
        cond = (torch.randn(rc.B, 2 * H * n, generator=g) * 0.3).to(dev())
    elif mode == "per_row":
        cond = (torch.randn(rc.R, 2 * H * n, generator=g) * 0.3).to(dev())
    res = []
    for fused in (True, False):
        wn.set_fused(fused)                                          # fragment-ordered vs row-major weight images
        modules.prepare_all(wn)
        out, saved = flow_impl.wn_fwd(rc, wn, h0, cond, True, 77, cond_per_row=mode == "per_row")
        with wgrad.WgradQueue(dev(), site=wn):
            dh0, grads, dcond = flow_impl.wn_bwd(rc, wn, saved, dskip, want_dcond=cond is not None, cond_per_row=mode == "per_row")
        torch.cuda.synchronize()
        res.append((out.float(), [t.float() for t in saved[1]], [s.float() for s in saved[2]], saved[3].float(), dh0.float(),
                    None if dcond is None else dcond.clone(), {id(k): v.float().clone() for k, v in grads.items()}))
    (o1, t1, s1, a1, d1, c1, g1), (o2, t2, s2, a2, d2, c2, g2) = res
    valid = rc.rowmask.bool()
    assert relerr(o1[valid], o2[valid]) < 1e-2
    for u, v in zip(t1 + s1, t2 + s2):
        assert (u[valid] - v[valid]).abs().max().item() < 2e-2          # bf16 x_i differ by an ulp here and there -> gate inputs move
    assert relerr(a1[valid], a2[valid]) < 2e-2
    assert relerr(d1[valid], d2[valid]) < 2e-2, relerr(d1[valid], d2[valid])
    if c1 is not None:
        assert relerr(c1, c2) < 2e-2, relerr(c1, c2)
    assert g1.keys() == g2.keys()
    for k in g1:
        assert relerr(g1[k], g2[k]) < 2e-2, relerr(g1[k], g2[k])


@pytest.mark.parametrize("speaker,sigmoid_scale", [(False, False), (True, False), (False, True)])
def test_fused_boundary_kernels_match_the_five_kernel_path(built, speaker, sigmoid_scale):
    """csrc/wn_boundary.hip (skip GEMM + end conv + coupling + ActNorm + InvConvNear + start conv as ONE kernel between two
    WaveNets, and the mirror kernel in the backward) against round 1's launch sequence on the same module, inputs and
    dropout seeds (train mode): z, log-det, input gradient, speaker-vector gradient and every parameter gradient; three
    blocks (head-only, head+tail, tail-only variants all run), ragged lengths incl. a 2-frame utterance, row count not a
    multiple of 64."""
    from glow_tts_amd import models
    gin = 256 if speaker else 0
    dec = fill_module(models.FlowSpecDecoder(80, 192, 5, 1, 3, 4, p_dropout=0.05, sigmoid_scale=sigmoid_scale, gin_channels=gin),
                      "decoder.").to(dev()).train()
    assert dec.fused_boundary
    lens = [140, 66, 2, 128, 90]
    B, T = len(lens), 140
    m = lens_mask(lens, T).to(dev())
    gen = torch.Generator().manual_seed(21)
    y0 = (torch.randn(B, 80, T, generator=gen)).to(dev()) * m
    g0 = torch.randn(B, gin, 1, generator=gen).to(dev()) if speaker else None
    rz = (torch.randn(B, 80, T, generator=gen)).to(dev()) * m
    rl = (torch.randn(B, generator=gen) * 0.1).to(dev())
    res = []
    for fused in (True, False):
        assert dec.set_fused_boundary(fused) == fused
        dec._step = 5                                             # same dropout seeds in both runs
        for p in dec.parameters():
            p.grad = None
        y = y0.clone().requires_grad_(True)
        g = g0.clone().requires_grad_(True) if speaker else None
        z, ld = dec(y, m, g=g)
        ((z * rz).sum() + (ld * rl).sum()).backward()
        torch.cuda.synchronize()
        res.append((z.detach().clone(), ld.detach().clone(), y.grad.clone(), None if g is None else g.grad.clone(),
                    {n: p.grad.clone() for n, p in dec.named_parameters()}))
    (z1, l1, d1, c1, g1), (z2, l2, d2, c2, g2) = res
    assert relerr(z1, z2) < 1e-2, relerr(z1, z2)
    assert (l1 - l2).abs().max().item() < 1e-3 * max(1.0, l2.abs().max().item()), (l1, l2)
    assert relerr(d1, d2) < 2e-2, relerr(d1, d2)
    if speaker:
        assert relerr(c1, c2) < 2e-2, relerr(c1, c2)
    worst = ("", 0.0)
    for n in g1:
        e = relerr(g1[n], g2[n])
        if e > worst[1]:
            worst = (n, e)
        assert e < 2e-2, (n, e)
    print("fused boundary vs five kernels: worst parameter gradient", worst)


@pytest.mark.parametrize("mode", ["none", "speaker", "per_row"])
def test_wn_stack_kernel_is_bit_identical_to_the_per_layer_kernels(built, mode):
    """csrc/wn_stack.hip (all four layers of a WaveNet in one launch, the 2-row halo between layers recomputed per 52-row tile)
    against four gt_wn_layer_fwd launches: same arithmetic in the same order and the same dropout hash, so T / S / acts / x_i
    must be EQUAL on every valid row — ragged rows, utterances of 1 and 2 frames, a row count that is neither a multiple of
    52 nor of 64, train mode (p = 0.05)."""
    from glow_tts_amd import flow_impl, modules, ops
    H, n = 192, 4
    gin = 256 if mode == "speaker" else 0
    wn = fill_module(modules.WN(160, H, 5, 1, n, gin, 0.05), "wn.").to(dev())
    modules.prepare_all(wn)
    lens = [131, 70, 2, 1, 64, 97]
    lt = torch.tensor(lens, dtype=torch.int32, device=dev())
    rc = ops.RowsCtx(lt, 131, lengths_host=lens, round_to=8)
    assert rc.R % 64 != 0 and rc.R % 52 != 0
    g = torch.Generator().manual_seed(31)
    h0 = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    cond = None
    if mode == "speaker":
        cond = (torch.randn(rc.B, 2 * H * n, generator=g) * 0.3).to(dev())
    elif mode == "per_row":
        cond = (torch.randn(rc.R, 2 * H * n, generator=g) * 0.3).to(dev())
    res = []
    for stack in (True, False):
        wn.set_stack(stack, stack)
        try:
            out, (xs, ts, ss, acts_all, p, seed) = flow_impl.wn_fwd(rc, wn, h0, cond, True, 123, cond_per_row=mode == "per_row")
        finally:
            wn.set_stack(True, True)
        torch.cuda.synchronize()
        res.append((out, xs, ts, ss, acts_all))
    valid = rc.rowmask.bool()
    (o1, x1, t1, s1, a1), (o2, x2, t2, s2, a2) = res
    # every row m < R is owned by exactly one tile: acts / T / S are written for masked rows too, as the per-layer kernels do
    assert torch.equal(a1, a2)
    for u, v in zip(t1 + s1 + x1, t2 + s2 + x2):
        assert torch.equal(u, v)
    assert torch.equal(o1[valid], o2[valid])


@pytest.mark.parametrize("mode", ["none", "speaker", "per_row"])
def test_wn_stack_backward_is_bit_identical_to_the_per_layer_kernels(built, mode):
    """gt_wn_stack_bwd (the WaveNet's whole data-gradient chain in one launch, halo recomputed) against gt_gate_bwd + three
    gt_wn_layer_bwd + the bottom data gradient: d h0, the conditioning gradient and every parameter gradient EQUAL (the batched
    weight-gradient kernels read the d pre / dX rows both paths wrote), dropout replayed, ragged rows."""
    from glow_tts_amd import flow_impl, modules, ops, wgrad
    H, n = 192, 4
    gin = 256 if mode == "speaker" else 0
    wn = fill_module(modules.WN(160, H, 5, 1, n, gin, 0.05), "wn.").to(dev())
    modules.prepare_all(wn)
    lens = [131, 70, 2, 1, 64, 97]
    lt = torch.tensor(lens, dtype=torch.int32, device=dev())
    rc = ops.RowsCtx(lt, 131, lengths_host=lens, round_to=8)
    g = torch.Generator().manual_seed(32)
    h0 = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    dskip = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    cond = None
    if mode == "speaker":
        cond = (torch.randn(rc.B, 2 * H * n, generator=g) * 0.3).to(dev())
    elif mode == "per_row":
        cond = (torch.randn(rc.R, 2 * H * n, generator=g) * 0.3).to(dev())
    res = []
    for stack in (True, False):
        wn.set_stack(stack, stack)
        try:
            out, saved = flow_impl.wn_fwd(rc, wn, h0, cond, True, 55, cond_per_row=mode == "per_row")
            with wgrad.WgradQueue(dev(), site=wn):
                dh0, grads, dcond = flow_impl.wn_bwd(rc, wn, saved, dskip, want_dcond=cond is not None, cond_per_row=mode == "per_row")
        finally:
            wn.set_stack(True, True)
        torch.cuda.synchronize()
        res.append((dh0.clone(), None if dcond is None else dcond.clone(), {id(k): v.clone() for k, v in grads.items()}))
    (d1, c1, g1), (d2, c2, g2) = res
    assert torch.equal(d1, d2)
    if c1 is not None:
        assert torch.equal(c1, c2)
    assert g1.keys() == g2.keys()
    for k in g1:
        assert torch.equal(g1[k], g2[k])


@pytest.mark.parametrize("n,lens", [(2, [3]), (3, [40, 1, 2]), (1, [70, 5]), (4, [2])])
def test_wn_stack_kernels_other_depths_and_tiny_batches(built, n, lens):
    """gt_wn_stack_fwd / _bwd with 1..4 layers (the rows a workgroup owns: 64 - 4 (n - 1)) and row counts below one tile:
    equal to the per-layer kernels, forward and backward."""
    from glow_tts_amd import flow_impl, modules, ops, wgrad
    H = 192
    wn = fill_module(modules.WN(160, H, 5, 1, n, 0, 0.05), "wn.").to(dev())
    modules.prepare_all(wn)
    T = max(lens)
    rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev()), T, lengths_host=lens, round_to=8)
    g = torch.Generator().manual_seed(7)
    h0 = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    dskip = ((torch.randn(rc.R, H, generator=g)).to(dev()) * rc.rowmask[:, None]).to(torch.bfloat16)
    res = []
    for stack in (True, False):
        wn.set_stack(stack, stack)
        try:
            out, saved = flow_impl.wn_fwd(rc, wn, h0, None, True, 9)
            with wgrad.WgradQueue(dev(), site=wn):
                dh0, grads, _ = flow_impl.wn_bwd(rc, wn, saved, dskip)
        finally:
            wn.set_stack(True, True)
        torch.cuda.synchronize()
        res.append((saved[3].clone(), [t.clone() for t in saved[1] + saved[2] + saved[0]], dh0.clone(), {id(k): v.clone() for k, v in grads.items()}))
    (a1, l1, d1, g1), (a2, l2, d2, g2) = res
    assert torch.equal(a1, a2) and torch.equal(d1, d2)
    for u, v in zip(l1, l2):
        assert torch.equal(u, v)
    for k in g1:
        assert torch.equal(g1[k], g2[k])
