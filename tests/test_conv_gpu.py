"""GPU tests: bf16 MFMA implicit-GEMM conv (gt_conv_gemm_bf16) vs plain PyTorch fp32 conv1d on the
same bf16-rounded operands.  Tolerance: fp32 accumulation of bf16 products -> 2e-3 relative to the
output scale (bf16 output rounding adds 2^-8)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def ref_conv(x_bct, w, b, pad):
    return F.conv1d(x_bct.float(), w.float(), None if b is None else b.float(), padding=pad)


def make(B, T, Cin, Cout, k, seed, lens=None):
    from glow_tts_amd import ops
    g = torch.Generator().manual_seed(seed)
    lens = lens or [T] * B
    ctx = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev()), T)
    x = torch.randn(B, Cin, T, generator=g).to(dev())
    x = x * ctx.rowmask2d[:, ops.HALO:ops.HALO + T].unsqueeze(1)
    w = (torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5).to(dev())
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev())
    return ctx, x, w, b


@pytest.mark.parametrize("Cin,Cout,k", [(192, 384, 5), (80, 192, 1), (192, 160, 1), (192, 768, 3), (768, 192, 3),
                                        (192, 192, 1), (192, 80, 1), (384, 192, 1)])
def test_conv_fwd_matches_torch(built, Cin, Cout, k):
    from glow_tts_amd import ops
    B, T = 3, 150
    ctx, x, w, b = make(B, T, Cin, Cout, k, seed=Cin + Cout + k, lens=[150, 97, 1])
    xb = x.to(torch.bfloat16)
    wb = w.to(torch.bfloat16)
    pc = ops.PackedConv(Cout, Cin, k).pack(wb.float())
    xr = ctx.to_rows(xb)
    y = ops.conv_rows(xr, pc, ctx, bias=b, out_f32=True)
    got = ctx.from_rows(y)
    want = ref_conv(xb, wb, b, k // 2)
    scale = want.abs().max().item()
    assert torch.allclose(got, want, atol=2e-3 * scale, rtol=0), (got - want).abs().max().item() / scale
    # masked + relu + bf16 out
    y2 = ops.conv_rows(xr, pc, ctx, bias=b, relu=True, mask=True)
    got2 = ctx.from_rows(y2).float()
    want2 = torch.relu(want) * ctx.rowmask2d[:, ops.HALO:ops.HALO + T].unsqueeze(1)
    assert torch.allclose(got2, want2, atol=1e-2 * scale, rtol=0)
    # halo rows of a masked output are exactly zero
    assert y2.reshape(B, ctx.Tp, Cout)[:, :ops.HALO].abs().max().item() == 0
    assert y2.reshape(B, ctx.Tp, Cout)[:, ops.HALO + T:].abs().max().item() == 0


def test_conv_dgrad_and_addend(built):
    """Data gradient of a k=5 conv = conv of dY with the dgrad-packed weights; epilogue addend."""
    from glow_tts_amd import ops
    B, T, Cin, Cout, k = 2, 96, 192, 384, 5
    ctx, x, w, b = make(B, T, Cin, Cout, k, seed=7)
    wb = w.to(torch.bfloat16)
    pc = ops.PackedConv(Cout, Cin, k).pack(wb.float())
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(B, Cout, T, generator=g).to(dev()).to(torch.bfloat16)
    xx = x.clone().float().requires_grad_(True)
    F.conv1d(xx, wb.float(), None, padding=2).backward(dy.float())
    add = torch.randn(B, Cin, T, generator=g).to(dev()).to(torch.bfloat16)
    dx = ops.conv_rows(ctx.to_rows(dy), pc, ctx, dgrad=True, addend=ctx.to_rows(add), mask=True)
    got = ctx.from_rows(dx).float()
    want = xx.grad + add.float()
    scale = want.abs().max().item()
    assert torch.allclose(got, want, atol=1e-2 * scale, rtol=0), (got - want).abs().max().item() / scale


def test_conv_weight_norm_pack(built):
    """w = g * v / ||v|| (torch weight_norm dim 0) folded into the packing."""
    from glow_tts_amd import ops
    B, T, Cin, Cout, k = 2, 64, 192, 192, 1
    ctx, x, v, b = make(B, T, Cin, Cout, k, seed=11)
    gw = (torch.rand(Cout, 1, 1, generator=torch.Generator().manual_seed(5)) + 0.5).to(dev())
    w = gw * v / v.reshape(Cout, -1).norm(dim=1).reshape(Cout, 1, 1)
    pc = ops.PackedConv(Cout, Cin, k).pack(v, gw)
    assert torch.allclose(pc.inv_norm, 1.0 / v.reshape(Cout, -1).norm(dim=1), rtol=1e-5)
    xb = x.to(torch.bfloat16)
    y = ops.conv_rows(ctx.to_rows(xb), pc, ctx, bias=b, out_f32=True)
    want = ref_conv(xb, w.to(torch.bfloat16), b, 0)
    scale = want.abs().max().item()
    assert torch.allclose(ctx.from_rows(y), want, atol=3e-3 * scale, rtol=0)


def test_conv_gate_epilogue(built):
    """in_layer + cond + tanh*sigmoid gate (modules.py:152-163, commons.py:61-68), no dropout."""
    from glow_tts_amd import ops
    B, T, H, k = 2, 80, 192, 5
    ctx, x, w, b = make(B, T, H, 2 * H, k, seed=21, lens=[80, 33])
    wb = w.to(torch.bfloat16)
    pc = ops.PackedConv(2 * H, H, k, gate=True).pack(wb.float())
    cond = torch.randn(B, 2 * H, generator=torch.Generator().manual_seed(9)).to(dev())
    xb = x.to(torch.bfloat16)
    acts, t, s = ops.conv_rows(ctx.to_rows(xb), pc, ctx, bias=b, cond=cond, gate=True)
    pre = ref_conv(xb, wb, b, 2) + cond.unsqueeze(-1)
    want_t, want_s = torch.tanh(pre[:, :H]), torch.sigmoid(pre[:, H:])
    assert torch.allclose(ctx.from_rows(t).float(), want_t, atol=1.5e-2)
    assert torch.allclose(ctx.from_rows(s).float(), want_s, atol=1.5e-2)
    assert torch.allclose(ctx.from_rows(acts).float(), want_t * want_s, atol=1.5e-2)


def test_conv_gate_dropout_statistics(built):
    from glow_tts_amd import ops
    B, T, H, k = 2, 128, 192, 5
    ctx, x, w, b = make(B, T, H, 2 * H, k, seed=22)
    pc = ops.PackedConv(2 * H, H, k, gate=True).pack(w)
    xr = ctx.to_rows(x.to(torch.bfloat16))
    a0, t0, _ = ops.conv_rows(xr, pc, ctx, bias=None, gate=True)
    a1, t1, _ = ops.conv_rows(xr, pc, ctx, bias=None, gate=True, drop_p=0.25, seed=1234)
    a2, t2, _ = ops.conv_rows(xr, pc, ctx, bias=None, gate=True, drop_p=0.25, seed=1234)
    assert torch.equal(a1, a2)                                   # replayable
    dropped = (t1.float() == 0) & (t0.float().abs() > 1e-3)      # tanh(0) = 0 where the pre-activation was dropped
    frac = dropped.float().mean().item()
    assert 0.2 < frac < 0.3, frac


def test_gate_backward_epilogue_replays_forward_dropout(built):
    """gate == 2 (gate backward fused behind a dgrad GEMM): against the standalone gt_gate_bwd on the same
    d acts, and the dropout mask it replays is exactly the one the forward gate conv drew (same seed):
    with zero bias a dropped pre-activation gives T == 0 / S == 0.5 in the forward and a zero gradient here."""
    from glow_tts_amd import _lib, ops
    B, T, H, k, p, seed = 2, 96, 192, 5, 0.3, 1234
    ctx, x, w, b = make(B, T, H, 2 * H, k, seed=11)
    g = torch.Generator().manual_seed(5)
    R = ctx.R
    xr = ctx.to_rows(x.to(torch.bfloat16))
    pc_in = ops.PackedConv(2 * H, H, k, gate=True).pack(w)
    _, t, s = ops.conv_rows(xr, pc_in, ctx, gate=True, drop_p=p, seed=seed)                 # no bias: dropped -> pre == 0
    valid = ctx.rowmask.bool()
    keep_t, keep_s = (t.float() != 0)[valid], (s.float() != 0.5)[valid]
    assert abs(keep_t.float().mean().item() - (1 - p)) < 0.02 and abs(keep_s.float().mean().item() - (1 - p)) < 0.02
    # d acts = dres @ W_res + via_skip
    wres = (torch.randn(H, H, 1, generator=g) / H ** 0.5).to(dev())
    pc_res = ops.PackedConv(H, H, 1).pack(wres)
    dres = (torch.randn(R, H, generator=g).to(dev()) * ctx.rowmask[:, None]).to(torch.bfloat16)
    via = (torch.randn(R, H, generator=g).to(dev()) * ctx.rowmask[:, None]).to(torch.bfloat16)
    fused = ops.conv_rows(dres, pc_res, ctx, dgrad=True, addend=via, gate=2, gate_t=t, gate_s=s, drop_p=p, seed=seed)
    dacts = ops.conv_rows(dres, pc_res, ctx, dgrad=True, addend=via)
    want = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev())
    L = _lib.lib()
    _lib.check(L.gt_gate_bwd(_lib.ptr(dacts), H, _lib.ptr(t), _lib.ptr(s), H, _lib.ptr(want), 2 * H, None, R, H, float(p), seed,
                             _lib.ptr(ops.seed_word(dev())), _lib.current_stream(dev())), "gt_gate_bwd")
    torch.cuda.synchronize()
    assert fused.shape == (R, 2 * H)
    scale = want.float().abs().max().item()
    assert torch.allclose(fused.float()[valid], want.float()[valid], atol=2e-2 * scale, rtol=0)   # d acts is bf16-rounded in `want`
    nz_t, nz_s = (fused[:, :H].float() != 0)[valid], (fused[:, H:].float() != 0)[valid]
    # a kept element can still be zero only if d acts or a saturated gate factor is zero: compare on the forward's dropped set
    assert not (nz_t & ~keep_t).any()
    assert (nz_t | ~keep_t).float().mean().item() > 0.98
    # sigmoid half: d pre_s = d*T*S*(1-S) also vanishes where T was dropped, and a KEPT pre-activation below 2^-8
    # rounds to S == 0.5 in bf16: compare where T survived and allow those few
    assert (nz_s & ~keep_s).float().mean().item() < 0.01 and (~nz_s & keep_s & keep_t).float().mean().item() < 0.02


TILES = {"64x64": 1, "64x128": 2, "128x64": 3, "128x128": 4, "256x64": 5, "64x64taps": 6}


@pytest.mark.parametrize("Cin,Cout,k,tile,big", [
    (192, 192, 1, "64x64", False), (192, 192, 1, "128x64", False), (192, 768, 3, "64x128", False), (192, 768, 3, "128x128", False),
    (384, 192, 5, "128x64", False), (192, 160, 1, "128x64", False), (80, 192, 1, "64x64", False),
    # all taps of a K slice per stage (the text encoder's short, deep convs): k = 3 / k = 5 / k = 1, ragged channel counts
    (768, 192, 3, "64x64taps", False), (192, 768, 3, "64x64taps", False), (192, 192, 5, "64x64taps", False), (80, 160, 1, "64x64taps", False),
    # R >= 11k rows: what the library's own choice runs at cfg 2's skip GEMM and at every larger batch
    (192, 192, 1, "auto", True), (768, 192, 1, "auto", True), (192, 768, 3, "auto", True), (192, 192, 5, "128x64", True)])
def test_conv_every_tile_variant_matches_torch(built, Cin, Cout, k, tile, big):
    """Forward + data gradient of the implicit-GEMM conv with the tile FORCED through the C-ABI's `tile` argument (and the
    library's own choice at R >= 11k rows, where it picks the 128-row tiles), ragged lengths, vs torch conv1d."""
    from glow_tts_amd import ops
    B, T = (30, 400) if big else (3, 150)
    lens = ([400, 397, 1] + [380 - 3 * i for i in range(27)]) if big else [150, 97, 1]
    ctx, x, w, b = make(B, T, Cin, Cout, k, seed=Cin + Cout + k, lens=lens)
    assert not big or ctx.R >= 11000
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    pc = ops.PackedConv(Cout, Cin, k).pack(wb.float())
    t = TILES.get(tile, 0)
    if t in (2, 4) and pc.Np_f % 128:
        pytest.skip("128-column tiles need Np % 128 == 0")
    xr = ctx.to_rows(xb)
    y = ops.conv_rows(xr, pc, ctx, bias=b, out_f32=True, tile=t)
    want = ref_conv(xb, wb, b, k // 2)
    scale = want.abs().max().item()
    got = ctx.from_rows(y)
    assert torch.allclose(got, want, atol=2e-3 * scale, rtol=0), (got - want).abs().max().item() / scale
    g = torch.Generator().manual_seed(1)
    msk = ctx.rowmask2d[:, ops.HALO:ops.HALO + T].unsqueeze(1)
    dy = (torch.randn(B, Cout, T, generator=g).to(dev()) * msk).to(torch.bfloat16)
    xx = xb.float().requires_grad_(True)
    F.conv1d(xx, wb.float(), None, padding=k // 2).backward(dy.float())
    td = t if (t not in (2, 4) or pc.Np_d % 128 == 0) else 0
    dx = ctx.from_rows(ops.conv_rows(ctx.to_rows(dy), pc, ctx, dgrad=True, out_f32=True, tile=td))
    valid = msk.bool().expand_as(dx)
    s2 = xx.grad.abs().max().item()
    assert torch.allclose(dx[valid], xx.grad[valid], atol=2e-3 * s2, rtol=0), (dx - xx.grad)[valid].abs().max().item() / s2


@pytest.mark.parametrize("tile,big", [("64x128", False), ("128x128", False), ("256x64", False), ("auto", True)])
def test_gate_conv_every_tile_variant(built, tile, big):
    """WaveNet gate epilogue (tanh * sigmoid, T / S saved) on every gate tile variant vs torch, incl. R >= 11k rows."""
    from glow_tts_amd import ops
    H, k = 192, 5
    B, T = (30, 400) if big else (2, 96)
    lens = ([400, 397, 1] + [380 - 3 * i for i in range(27)]) if big else [96, 51]
    ctx, x, w, b = make(B, T, H, 2 * H, k, seed=3, lens=lens)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    pc = ops.PackedConv(2 * H, H, k, gate=True).pack(wb.float())
    a, t_, s_ = ops.conv_rows(ctx.to_rows(xb), pc, ctx, bias=b, gate=True, tile=TILES.get(tile, 0))
    pre = ref_conv(xb, wb, b, k // 2)
    want = torch.tanh(pre[:, :H]) * torch.sigmoid(pre[:, H:])
    valid = ctx.rowmask.bool()
    for got, ref in ((a, want), (t_, torch.tanh(pre[:, :H])), (s_, torch.sigmoid(pre[:, H:]))):
        assert torch.allclose(got.float()[valid], ctx.to_rows(ref)[valid], atol=1.2e-2, rtol=0)


def test_conv_tile_argument_is_validated(built):
    """A tile the shape does not allow is refused (GT_E_INVAL), not silently replaced."""
    from glow_tts_amd import ops
    ctx, x, w, b = make(2, 32, 192, 192, 1, seed=5)
    pc = ops.PackedConv(192, 192, 1).pack(w)
    xr = ctx.to_rows(x.to(torch.bfloat16))
    with pytest.raises(RuntimeError, match="GT_E_INVAL"):
        ops.conv_rows(xr, pc, ctx, tile=TILES["64x128"])        # Np = 192 is not a multiple of 128
    with pytest.raises(RuntimeError, match="GT_E_INVAL"):
        ops.conv_rows(xr, pc, ctx, tile=TILES["256x64"])        # gate-only variant


@pytest.mark.parametrize("tile,big", [("64x64", False), ("64x128", False), ("64x64taps", False), ("128x128", True), ("auto", True)])
def test_relu_dropout_backward_in_the_data_gradient_epilogue(built, tile, big):
    """gate == 3: the backward of y = dropout(relu(conv_1(x))) (attentions.py:368-370) rides in the epilogue of conv_2's data-gradient
    GEMM — d c1 = d f1 / (1 - p) where the saved y is non-zero, else 0 — on every tile the text encoder's shapes select (64-row tiles
    at cfg 2, the 128 x 128 tile at cfg 3's row counts), against the unfused pair (data gradient, then gt_relu_drop_bwd)."""
    from glow_tts_amd import _lib, ops
    B, T = (30, 400) if big else (3, 150)
    lens = ([400, 397, 1] + [380 - 3 * i for i in range(27)]) if big else [150, 97, 1]
    F_, C, k, p = 768, 192, 3, 0.1
    ctx, x, w, b = make(B, T, F_, C, k, seed=77, lens=lens)          # conv_2: F_ -> C; its data gradient maps [R, C] -> [R, F_]
    pc = ops.PackedConv(C, F_, k).pack(w.to(torch.bfloat16).float())
    g = torch.Generator().manual_seed(3)
    R = ctx.R
    dy = (torch.randn(R, C, generator=g).to(dev()) * ctx.rowmask[:, None]).to(torch.bfloat16)
    y = torch.relu(torch.randn(R, F_, generator=g)).to(dev())
    y = (y * (torch.rand(R, F_, generator=g).to(dev()) > p)).to(torch.bfloat16)       # ~55 % zeros: relu'd or dropped
    t = TILES.get(tile, 0)
    want_d = ops.conv_rows(dy, pc, ctx, dgrad=True, tile=0 if t == 6 else t)
    want = torch.empty_like(want_d)
    _lib.check(_lib.lib().gt_relu_drop_bwd(_lib.ptr(want_d), F_, _lib.ptr(y), F_, _lib.ptr(want), F_, R, F_, p, _lib.current_stream(dev())),
               "gt_relu_drop_bwd")
    got = ops.conv_rows(dy, pc, ctx, dgrad=True, gate=3, gate_t=y, drop_p=p, tile=t)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert ((got != 0) <= (y != 0)).all()                              # nothing leaks through a zero of the saved activation
    # the fused form scales the fp32 accumulator, the unfused one the bf16-rounded data gradient: one bf16 ulp apart at most
    scale = want.float().abs().max().item()
    assert torch.allclose(got.float(), want.float(), atol=1e-2 * scale, rtol=2e-2), (got.float() - want.float()).abs().max().item() / scale
