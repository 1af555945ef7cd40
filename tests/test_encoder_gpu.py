"""GPU parity tests: text encoder, rel-pos attention, logp, prior expansion, mle loss and the whole
training forward/backward (glow_tts_amd, HIP through the C-ABI) vs the float oracle
(oracle/glowtts_ref.py, pinned to the reference by tests/golden/float_golden.npz).

Tolerances: bf16 GEMM operands / bf16 hidden activations -> 3e-2 of max-abs on activations,
6e-2 on parameter gradients; fp32-only kernels (logp, expansion, loss) 1e-4."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from fill import fill_module  # noqa: E402
from oracle import glowtts_ref as R  # noqa: E402
from oracle import mas as omas  # noqa: E402

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def relerr(a, b):
    return (a - b).abs().max().item() / max(1e-6, b.abs().max().item())


def grad_ok(a, b, tol, atol=2e-3, name=""):
    """Gradient agreement in relative L2 norm (plus a loose max-abs bound).  Max-abs alone is the wrong
    yardstick downstream of a ReLU: a pre-activation within bf16 rounding of zero switches the unit
    on/off relative to the fp32 oracle, which moves single elements by their full magnitude while the
    tensor as a whole stays within tolerance.
    The key bias of a softmax attention has a mathematically zero gradient (a constant added to every
    key cancels in the softmax): both sides hold rounding noise only."""
    if name.endswith("conv_k.bias"):
        return a.abs().max().item() < 5e-2 and b.abs().max().item() < 1e-4
    l2 = (a - b).norm().item() / max(1e-12, b.norm().item())
    mx = (a - b).abs().max().item()
    return l2 <= tol and mx <= 0.5 * b.abs().max().item() + atol


def lens_mask(lengths, T):
    l = torch.tensor(lengths)
    return (torch.arange(T)[None, :] < l[:, None]).unsqueeze(1).float()


def cpu_state(mod, prefix=""):
    P = {prefix + k: v.detach().cpu().float().clone() for k, v in mod.state_dict().items()}
    for v in P.values():
        v.requires_grad_(True)
    return P


# MFMA kernels: <= 160 (5 key tiles in registers), <= 256 (8 tiles; backward one tile at a time), <= 384 (12 tiles, one
# operand from L2; cfg3: T_x <= 375); above that the generic kernels (400)
@pytest.mark.parametrize("T", [3, 5, 37, 150, 161, 200, 256, 257, 300, 375, 384, 400])
def test_mha_fwd_bwd(built, T):
    from glow_tts_amd import attentions
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
    att = att.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    od = att(xd, xd, am.to(dev()))
    valid = xm.bool().expand_as(o)
    assert relerr(od.detach().cpu()[valid], o.detach()[valid]) < 3e-2
    pv = (xm.unsqueeze(-1) * xm.unsqueeze(2)).bool().expand_as(p)
    assert (att.attn.cpu()[pv] - p.detach()[pv]).abs().max() < 2e-2
    (od * r.to(dev())).sum().backward()
    assert relerr(xd.grad.cpu(), xx.grad) < 4e-2
    for name, prm in att.named_parameters():
        assert grad_ok(prm.grad.cpu(), P["mha." + name].grad, 6e-2, name=name), name


def test_encoder_stack_fwd_bwd(built):
    from glow_tts_amd import attentions
    enc = fill_module(attentions.Encoder(192, 768, 2, 2, 3, 0.1, window_size=4), "enc.").eval()
    P = cpu_state(enc, "enc.")
    T, lens = 41, [41, 17, 30]
    xm = lens_mask(lens, T)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(3, 192, T, generator=g) * xm
    xx = x.clone().requires_grad_(True)
    o = R.encoder_fwd(P, "enc.", xx, xm, n_layers=2)
    r = torch.randn(o.shape, generator=g)
    (o * r).sum().backward()
    enc = enc.to(dev())
    xd = x.to(dev()).requires_grad_(True)
    od = enc(xd, xm.to(dev()))
    assert relerr(od.detach().cpu(), o.detach()) < 3e-2
    (od * r.to(dev())).sum().backward()
    assert relerr(xd.grad.cpu(), xx.grad) < 5e-2
    for name, prm in enc.named_parameters():
        assert grad_ok(prm.grad.cpu(), P["enc." + name].grad, 8e-2, name=name), name


def test_encoder_speaker_conditioning_fwd_bwd(built):
    """Encoder.cond_g (attentions.py:66-67): the speaker vector added before layer index 2 — output, input gradient, the
    gradient of g and of cond_g's parameters against the oracle (pinned to the reference by float_golden.npz: encg_*)."""
    from glow_tts_amd import attentions
    enc = fill_module(attentions.Encoder(192, 768, 2, 3, 3, 0.1, window_size=4, gin_channels=256), "enc.").eval()
    P = cpu_state(enc, "enc.")
    T, lens = 41, [41, 17, 30]
    xm = lens_mask(lens, T)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 192, T, generator=g) * xm
    spk = torch.randn(3, 256, 1, generator=g)
    xx, gg = x.clone().requires_grad_(True), spk.clone().requires_grad_(True)
    o = R.encoder_fwd(P, "enc.", xx, xm, g=gg, n_layers=3)
    o0 = R.encoder_fwd(P, "enc.", x, xm, g=None, n_layers=3)
    assert relerr(o0.detach(), o.detach()) > 0.05                      # the conditioning matters at these weights
    r = torch.randn(o.shape, generator=g)
    (o * r).sum().backward()
    enc = enc.to(dev())
    xd, gd = x.to(dev()).requires_grad_(True), spk.to(dev()).requires_grad_(True)
    od = enc(xd, xm.to(dev()), g=gd)
    assert relerr(od.detach().cpu(), o.detach()) < 3e-2
    (od * r.to(dev())).sum().backward()
    assert relerr(xd.grad.cpu(), xx.grad) < 5e-2
    assert grad_ok(gd.grad.cpu(), gg.grad, 8e-2), relerr(gd.grad.cpu(), gg.grad)
    for name, prm in enc.named_parameters():
        assert grad_ok(prm.grad.cpu(), P["enc." + name].grad, 8e-2, name=name), name


@pytest.mark.parametrize("ragged", [False, True])
def test_train_forward_backward_with_language_and_speaker(built, ragged):
    """cfg 5's text side: language id -> emb_l -> concatenated to the 188-channel token embedding at every position
    (models.py:654-664, 698-699, 1011-1012) and added through DurationPredictor.cond_lang; together with the speaker
    vector.  Whole training forward / backward against the oracle (language path pinned by tel_* / dpl_* goldens),
    including the gradients of emb_l, the narrower emb and cond_lang."""
    from glow_tts_amd import models, ops
    gen = _make_generator(3, 256, n_lang=3, lin_channels=4)
    P = cpu_state(gen)
    g = torch.Generator().manual_seed(17)
    B, Tx, Ty = 3, 33, 96
    xl, yl = torch.tensor([33, 20, 9]), torch.tensor([96, 60, 30])
    ids = torch.randint(1, 148, (B, Tx), generator=g) * (torch.arange(Tx)[None, :] < xl[:, None])
    y = torch.randn(B, 80, Ty, generator=g) * lens_mask(yl.tolist(), Ty)
    spk = torch.randn(B, 256, 1, generator=g)
    lid = torch.tensor([2, 0, 1])
    gen = gen.to(dev())
    gen.rows_cfg.ragged = ragged
    try:
        (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, l_length, _, _), _, _ = \
            gen(ids.to(dev()), xl.to(dev()), y.to(dev()), yl.to(dev()), g=spk.to(dev()), l=lid.to(dev()))
    finally:
        gen.rows_cfg.ragged = False
    l_mle = models.mle_loss(z, z_m, z_logs, logdet, z_mask)
    (l_mle + l_length.sum()).backward()
    lvec = torch.nn.functional.embedding(lid, P["emb_l.weight"]).unsqueeze(-1)
    hp = dict(HP, n_layers_enc=3)
    out = R.train_forward(P, ids, xl, y, yl, lambda logp, mask: attn.squeeze(1).cpu().float(), hp, g=spk, l=lvec)
    out["loss"].backward()
    assert relerr(x_m.detach().cpu(), out["x_m"].detach()) < 3e-2
    assert relerr(z_m.detach().cpu(), out["z_m"].detach()) < 3e-2
    assert relerr(l_length.detach().cpu(), out["l_length"].detach()) < 5e-2
    bad = []
    for name, prm in gen.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert prm.grad is None or prm.grad.abs().max().item() == 0, name
            continue
        assert prm.grad is not None, name
        # the duration predictor's input gradient only exists as the bf16 output of a data-gradient GEMM; cond / cond_lang
        # gradients are per-utterance sums of it times a 256- / 4-vector: 0.08-0.12 in relative L2
        tol = 0.15 if (".pre.conv_layers." in name or "proj_w.cond" in name) else (0.2 if "emb_rel_" in name else 0.1)
        if not grad_ok(prm.grad.cpu(), ref, tol, name=name):
            bad.append((name, round(relerr(prm.grad.cpu(), ref), 3)))
    assert not bad, bad
    for key in ("emb_l.weight", "encoder.emb.weight", "encoder.proj_w.cond_lang.weight"):
        assert dict(gen.named_parameters())[key].grad.abs().max().item() > 0, key


def test_logp_kernel(built):
    from glow_tts_amd.text_models import _LogpMasFn
    g = torch.Generator().manual_seed(4)
    B, C, Tx, Ty = 3, 80, 45, 130
    x_m = torch.randn(B, C, Tx, generator=g)
    x_logs = torch.randn(B, C, Tx, generator=g) * 0.2
    z = torch.randn(B, C, Ty, generator=g)
    xl = torch.tensor([45, 20, 1]); yl = torch.tensor([130, 64, 2])
    for mean_only in (True, False):
        want = R.logp_lattice(x_m, torch.zeros_like(x_m) if mean_only else x_logs, z)
        logp, mas = _LogpMasFn.run(x_m.to(dev()), x_logs.to(dev()), z.to(dev()), xl.to(dev()), yl.to(dev()), mean_only)
        assert relerr(logp.cpu(), want) < 1e-5
        # MAS on exactly this lattice is bit-exact with the oracle
        mask = (lens_mask(xl.tolist(), Tx).unsqueeze(-1) * lens_mask(yl.tolist(), Ty).unsqueeze(2)).squeeze(1)
        p = omas.oracle_maximum_path(logp.cpu().numpy(), mask.numpy())
        assert np.array_equal(mas.path.cpu().numpy().astype(np.int32), p)


HP = dict(hidden_channels=192, n_layers_enc=2, n_heads=2, window_size=4, kernel_size=3, prenet=True, mean_only=True,
          n_blocks_dec=2, n_block_layers=4, kernel_size_dec=5, n_sqz=2)


def _make_generator(n_layers_enc=2, gin_channels=0, with_prosody_wn=False, n_lang=0, lin_channels=0):
    from glow_tts_amd import models
    return fill_module(models.FlowGenerator(148, 192, 768, 256, 80, use_sdp=False, kernel_size=3, n_heads=2, n_layers_enc=n_layers_enc, p_dropout=0.1,
                                            n_blocks_dec=2, kernel_size_dec=5, dilation_rate=1, n_block_layers=4,
                                            p_dropout_dec=0.05, n_sqz=2, window_size=4, mean_only=True, prenet=True,
                                            gin_channels=gin_channels, with_prosody_wn=with_prosody_wn, n_lang=n_lang,
                                            lin_channels=lin_channels), "").eval()


def test_text_encoder_fwd(built):
    gen = _make_generator()
    P = cpu_state(gen)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(1, 148, (2, 23), generator=g); xl = torch.tensor([23, 9])
    x, x_m, x_logs, m = R.text_encoder_fwd(P, "encoder.", ids, xl, n_layers=2)
    gen = gen.to(dev())
    gen.prepare()
    xd, xmd, xld, md = gen.encoder(ids.to(dev()), xl.to(dev()), prepared=True)
    assert torch.equal(md.cpu(), m)
    assert relerr(xd.detach().cpu(), x.detach()) < 3e-2 and relerr(xmd.detach().cpu(), x_m.detach()) < 3e-2
    assert xld.abs().max().item() == 0


@pytest.mark.parametrize("Tx,Ty,xl,yl,ragged,gin", [(21, 64, [21, 12], [64, 37], False, 0), (21, 64, [21, 12], [64, 37], True, 0),
                                                     (300, 640, [300, 131], [640, 402], True, 0),       # cfg3-like (T_x > 256)
                                                     (300, 640, [300, 131], [640, 402], False, 0),
                                                     (45, 130, [45, 20], [130, 64], False, 256),        # cfg4-like: speaker vector g
                                                     (45, 130, [45, 20], [130, 64], True, 256),
                                                     (45, 130, [45, 20], [130, 64], True, -256)])       # cfg5-like: g + pitch + energy into the decoder
def test_train_forward_backward_vs_oracle(built, Tx, Ty, xl, yl, ragged, gin):
    """Whole hot path: TextEncoder -> decoder -> logp -> MAS -> losses, forward and backward.  The
    alignment is compared on the HIP path's own lattice (bit-exact), then injected into the oracle so
    that the remaining quantities are comparable.  Uniform and ragged rows layouts; short and cfg3-like lengths;
    with gin: the multi-speaker form (cfg 4), g [b,256,1] into the encoder (3 layers, so that cond_g is reached),
    the duration predictor and every coupling block."""
    from glow_tts_amd import models, ops
    prosody = gin < 0                       # negative gin: also the decoder's per-frame pitch / energy WaveNets (with 30 % unvoiced frames)
    gin = abs(gin)
    gen = _make_generator(3 if gin else 2, gin, with_prosody_wn=prosody)
    P = cpu_state(gen)
    g = torch.Generator().manual_seed(7)
    B = 2
    spk = torch.randn(B, gin, 1, generator=g) if gin else None
    pitch = energy = None
    if prosody:
        # raw contours whose LOG (models.py:1054-1071) is O(1), so that the closed-form cond_layer1 weights keep the decoder
        # in the numeric regime of the other cases (Hz-scale values saturate every gate and amplify z's bf16 error)
        pitch = torch.exp(torch.randn(B, 1, Ty, generator=g)) * (torch.rand(B, 1, Ty, generator=g) > 0.3)
        energy = torch.exp(0.5 * torch.randn(B, 1, Ty, generator=g))
    hp = dict(HP, n_layers_enc=3) if gin else HP
    ids = torch.randint(1, 148, (B, Tx), generator=g); xl = torch.tensor(xl)
    yl = torch.tensor(yl)
    y = torch.randn(B, 80, Ty, generator=g) * lens_mask(yl.tolist(), Ty)
    ids = ids * (torch.arange(Tx)[None, :] < xl[:, None])

    gen = gen.to(dev())
    gen.rows_cfg.ragged = ragged
    try:
        (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, l_length, _, _), _, _ = \
            gen(ids.to(dev()), xl.to(dev()), y.to(dev()), yl.to(dev()), g=None if spk is None else spk.to(dev()),
                pitch=None if pitch is None else pitch.to(dev()), energy=None if energy is None else energy.to(dev()))
    finally:
        gen.rows_cfg.ragged = False
    l_mle = models.mle_loss(z, z_m, z_logs, logdet, z_mask)
    loss = l_mle + l_length.sum()
    loss.backward()

    # alignment: bit-exact on the HIP lattice
    amask = (x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)).squeeze(1)
    p = omas.oracle_maximum_path(gen.last_logp.cpu().numpy(), amask.cpu().numpy())
    assert np.array_equal(attn.squeeze(1).cpu().numpy().astype(np.int32), p)

    out = R.train_forward(P, ids, xl, y, yl, lambda logp, mask: attn.squeeze(1).cpu().float(), hp, g=spk, pitch=pitch, energy=energy)
    out["loss"].backward()
    assert relerr(gen.last_logp.cpu(), out["logp"]) < 3e-2
    assert relerr(z.detach().cpu(), out["z"].detach()) < 3e-2
    assert relerr(z_m.detach().cpu(), out["z_m"].detach()) < 3e-2
    assert abs(l_mle.item() - out["l_mle"].item()) < 2e-2 * max(1.0, abs(out["l_mle"].item()))
    assert relerr(l_length.detach().cpu(), out["l_length"].detach()) < 5e-2
    worst, bad = [], []
    for name, prm in gen.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert prm.grad is None or prm.grad.abs().max().item() == 0, name
            continue
        assert prm.grad is not None, name
        if name.endswith("cond_layer1.weight_v"):           # one input channel: w = g*v/|v| — a mathematically zero gradient
            assert prm.grad.abs().max().item() <= 1e-3 * P[name[:-1] + "g"].grad.abs().max().item(), name
            continue
        e = relerr(prm.grad.cpu(), ref)
        worst.append((e, name))
        # prenet convs sit under three conv -> LayerNorm -> ReLU stages: bf16 ReLU flips (see grad_ok) compound, and their
        # relative-L2 error sits at 0.08-0.12 whatever the sequence length; everything else stays below 0.1
        # emb_rel_k / emb_rel_v gradients are sums of signed band entries over every (query, key) pair — heavy cancellation,
        # so bf16 noise weighs more (0.05-0.15 depending on the case)
        tol = 0.15 if ".pre.conv_layers." in name else (0.2 if "emb_rel_" in name else 0.1)
        if not grad_ok(prm.grad.cpu(), ref, tol, name=name):
            bad.append((name, round(e, 3)))
    worst.sort(reverse=True)
    print("worst grad errors:", worst[:5])
    assert not bad, bad


def test_full_size_cfg2_train_step_vs_oracle(built):
    """ONE full configs/base.json training step at the bench's size (B = 32, T_x <= 150, T_y <= 800, 6 encoder layers, 12 flow blocks,
    ragged rows: R_dec ~ 9 k, R_enc ~ 3.5 k — VERDICT r2: the full step was only ever benchmarked at this size): the MAS path is
    bit-exact on the product's own lattice, and with that path handed to the oracle the loss, z, the expanded prior and EVERY
    parameter gradient (encoder, duration predictor, decoder) agree with the fp32 oracle's full-batch step."""
    from glow_tts_amd import models
    gen = fill_module(models.FlowGenerator(148, 192, 768, 256, 80, use_sdp=False, kernel_size=3, n_heads=2, n_layers_enc=6, p_dropout=0.1,
                                           n_blocks_dec=12, kernel_size_dec=5, dilation_rate=1, n_block_layers=4, p_dropout_dec=0.05, n_sqz=2,
                                           window_size=4, mean_only=True, prenet=True), "").eval()
    P = cpu_state(gen)
    hp = dict(HP, n_layers_enc=6, n_blocks_dec=12)
    g = torch.Generator().manual_seed(1234)
    B, Tx, Ty = 32, 150, 800
    xl = torch.randint(60, Tx + 1, (B,), generator=g); xl[0] = Tx
    yl = torch.randint(150, Ty // 2 + 1, (B,), generator=g) * 2; yl[0] = Ty
    ids = torch.randint(1, 148, (B, Tx), generator=g) * (torch.arange(Tx)[None, :] < xl[:, None])
    y = torch.randn(B, 80, Ty, generator=g) * lens_mask(yl.tolist(), Ty)
    gen = gen.to(dev())
    gen.rows_cfg.ragged = True
    gen.rows_cfg.row_round = 512
    try:
        (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, l_length, _, _), _, _ = \
            gen(ids.to(dev()), xl.to(dev()), y.to(dev()), yl.to(dev()), lengths_host=(xl.tolist(), yl.tolist()))
    finally:
        gen.rows_cfg.ragged = False
    l_mle = models.mle_loss(z, z_m, z_logs, logdet, z_mask)
    loss = l_mle + l_length.sum()
    loss.backward()
    torch.cuda.synchronize()
    amask = (x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)).squeeze(1)
    p = omas.oracle_maximum_path(gen.last_logp.cpu().numpy(), amask.cpu().numpy())
    assert np.array_equal(attn.squeeze(1).cpu().numpy().astype(np.int32), p)                 # MAS: bit-exact at full size

    out = R.train_forward(P, ids, xl, y, yl, lambda logp, mask: attn.squeeze(1).cpu().float(), hp)
    out["loss"].backward()
    assert relerr(gen.last_logp.cpu(), out["logp"]) < 3e-2
    assert relerr(z.detach().cpu(), out["z"].detach()) < 3e-2
    assert relerr(z_m.detach().cpu(), out["z_m"].detach()) < 3e-2
    assert abs(l_mle.item() - out["l_mle"].item()) < 2e-2 * max(1.0, abs(out["l_mle"].item()))
    assert abs(loss.item() - out["loss"].item()) < 2e-2 * max(1.0, abs(out["loss"].item()))
    assert relerr(l_length.detach().cpu(), out["l_length"].detach()) < 5e-2
    worst, bad = [], []
    for name, prm in gen.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert prm.grad is None or prm.grad.abs().max().item() == 0, name
            continue
        assert prm.grad is not None, name
        e = relerr(prm.grad.cpu(), ref)
        worst.append((e, name))
        tol = 0.15 if ".pre.conv_layers." in name else (0.2 if "emb_rel_" in name else 0.1)
        if not grad_ok(prm.grad.cpu(), ref, tol, name=name):
            bad.append((name, round(e, 3)))
    worst.sort(reverse=True)
    print("full-size cfg 2 step, worst parameter-gradient errors:", worst[:5])
    assert len(worst) > 300 and not bad, bad


@pytest.mark.parametrize("T,ragged", [(150, False), (161, True), (200, False), (200, True), (256, True), (300, True), (384, False), (400, True)])
def test_attention_keeps_to_its_rows_in_merged_buffers(built, T, ragged):
    """q / k / v as windows of ONE [R, 3C] buffer and dq / dk / dv as windows of one gradient buffer (ld = 3C: how the encoder
    calls the kernels since round 1's `4715eb8`), each EXACTLY R rows long and embedded between guard rows: NaN guards around
    every input (a read outside the R rows poisons the output), canary guards around every output (a write outside them is
    seen), for every kernel variant by length (<= 160, <= 256, <= 384 MFMA; generic above), last utterance full length and,
    ragged, the last utterance owning the rounding rows.  This is the addressing audit of the round-1 memory fault
     as a test: results must equal the same call on separately allocated, generously padded buffers."""
    from glow_tts_amd import _lib, ops
    L = _lib.lib()
    H, D, C, win, GUARD = 2, 96, 192, 4, 8
    lens = [T - 7, 1, T]                                             # the LAST utterance is the longest
    lens_t = torch.tensor(lens, dtype=torch.int32, device=dev())
    rc = ops.RowsCtx(lens_t, T, lengths_host=lens, round_to=128) if ragged else ops.RowsCtx(lens_t, T)
    R_ = rc.R
    g = torch.Generator().manual_seed(T + int(ragged))
    qkv = (torch.randn(R_, 3 * C, generator=g) * 0.5).to(dev()) * rc.rowmask[:, None]
    do = (torch.randn(R_, C, generator=g)).to(dev()) * rc.rowmask[:, None]
    Ek = (torch.randn(2 * win + 1, D, generator=g) * 0.1).to(dev()); Ev = (torch.randn(2 * win + 1, D, generator=g) * 0.1).to(dev())
    st = _lib.current_stream(dev())
    row0 = _lib.ptr(rc.row0)

    def guarded(t, fill):
        """t [R, n] -> a [GUARD + R + GUARD, n] buffer with `fill` in the guard rows; returns (buffer, view of the middle)"""
        buf = torch.full((R_ + 2 * GUARD, t.shape[1]), fill, dtype=t.dtype, device=dev())
        buf[GUARD:GUARD + R_] = t
        return buf, buf[GUARD:GUARD + R_]

    def run(guard):
        nan, can = (float("nan"), 768.0) if guard else (0.0, 0.0)
        qb, qv = guarded(qkv.to(torch.bfloat16), nan)
        dob, dov = guarded(do.to(torch.bfloat16), nan)
        ob, ov = guarded(torch.zeros(R_, C, dtype=torch.bfloat16, device=dev()), can)
        db, dv_ = guarded(torch.zeros(R_, 3 * C, dtype=torch.bfloat16, device=dev()), can)
        P = torch.empty(rc.B, H, T, T, dtype=torch.float32, device=dev())
        q, k, v = qv[:, :C], qv[:, C:2 * C], qv[:, 2 * C:]
        _lib.check(L.gt_attn_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), 3 * C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths), _lib.ptr(ov), C,
                                 _lib.ptr(P), rc.B, T, rc.Tp, row0, H, D, win, 0.0, 0, None, st), "gt_attn_fwd")
        wsb = L.gt_attn_bwd_workspace_bytes(rc.B, T, H)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev())
        dEk, dEv = torch.zeros_like(Ek), torch.zeros_like(Ev)
        dq, dk, dvv = dv_[:, :C], dv_[:, C:2 * C], dv_[:, 2 * C:]
        _lib.check(L.gt_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), 3 * C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths), _lib.ptr(dov), C,
                                 _lib.ptr(P), _lib.ptr(ws), wsb, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dvv), 3 * C, _lib.ptr(dEk), _lib.ptr(dEv),
                                 rc.B, T, rc.Tp, row0, H, D, win, 0.0, 0, None, st), "gt_attn_bwd")
        torch.cuda.synchronize()
        return ob, db, dEk, dEv

    ob, db, dEk, dEv = run(True)
    for buf in (ob, db):                                             # canaries intact: nothing was written outside the R rows
        assert (buf[:GUARD].float() == 768.0).all() and (buf[GUARD + R_:].float() == 768.0).all()
    assert torch.isfinite(ob[GUARD:GUARD + R_].float()).all() and torch.isfinite(db[GUARD:GUARD + R_].float()).all()   # no NaN guard row was read
    assert torch.isfinite(dEk).all() and torch.isfinite(dEv).all()
    ob0, db0, dEk0, dEv0 = run(False)
    assert torch.equal(ob[GUARD:GUARD + R_], ob0[GUARD:GUARD + R_]) and torch.equal(db[GUARD:GUARD + R_], db0[GUARD:GUARD + R_])
    assert torch.allclose(dEk, dEk0, rtol=1e-3, atol=1e-4) and torch.allclose(dEv, dEv0, rtol=1e-3, atol=1e-4)
    assert db[GUARD:GUARD + R_].float().abs().max().item() > 0
