"""Forward/backward launch sequences of the text encoder on the rows layout (host side of the HIP
kernels).  Mirrors what autograd derives for the reference modules:

  mha_*        attentions.MultiHeadAttention     (attentions.py:231-272)
  layer_*      one attentions.Encoder layer      (attentions.py:69-84)
  crn_*        modules.ConvReluNorm (prenet)     (modules.py:95-102)
  dp_*         models.DurationPredictor          (models.py:585-612)

State convention: the residual stream is fp32 rows `x` (masked) with a bf16 masked copy `xb` that
feeds the MFMA GEMMs; gradients arrive as (dx_f32, dxb_bf16) pairs and are summed where the two
copies meet.
"""
import torch

from . import _lib
from .flow_impl import _st, conv_param_grads
from .ops import conv_rows, grad_accumulator, seed_word, zeros_small
from .ops import zeros_big as ops_zeros_big

LN_EPS = 1e-4


def _ln_fwd(rc, ln, a, y, p_in, seed_in, p_out, seed_out, relu, want_f32, C):
    L = _lib.lib()
    R = rc.R
    dev = rc.device
    out_f32 = torch.empty(R, C, dtype=torch.float32, device=dev) if want_f32 else None
    out_bf = torch.empty(R, C, dtype=torch.bfloat16, device=dev)
    mean = torch.empty(R, dtype=torch.float32, device=dev)
    rstd = torch.empty(R, dtype=torch.float32, device=dev)
    _lib.check(L.gt_layernorm_fwd(_lib.ptr(a), _lib.ptr(y), 0 if y is None else y.stride(0), _lib.ptr(ln.gamma), _lib.ptr(ln.beta),
                                  _lib.ptr(rc.rowmask), _lib.ptr(out_f32), _lib.ptr(out_bf), C, _lib.ptr(mean), _lib.ptr(rstd),
                                  R, C, LN_EPS, float(p_in), int(seed_in), float(p_out), int(seed_out), int(relu),
                                  _lib.ptr(seed_word(dev)) if (p_in > 0 or p_out > 0) else None, _st(dev)),
               "gt_layernorm_fwd")
    return out_f32, out_bf, (a, y, mean, rstd, p_in, seed_in, p_out, seed_out, relu, C)


def _ln_bwd(rc, ln, saved, dout_f32, dout_bf, want_da, want_dy, grads, relu_in=False):
    """relu_in: the bf16 input y is a ReLU's output (conv -> relu -> norm, models.py:591-598): dy is zeroed where y is zero, i.e.
    the ReLU's backward rides along instead of a launch of its own."""
    L = _lib.lib()
    a, y, mean, rstd, p_in, seed_in, p_out, seed_out, relu, C = saved
    R = rc.R
    dev = rc.device
    da = torch.empty(R, C, dtype=torch.float32, device=dev) if want_da else None
    dy = torch.empty(R, C, dtype=torch.bfloat16, device=dev) if want_dy else None
    dg = grad_accumulator(ln.gamma, (C,))
    db = grad_accumulator(ln.beta, (C,))
    from . import wgrad
    q = wgrad.active()
    if q is not None:
        # inside a module's backward (an open WgradQueue): per-workgroup partial sums now, ONE reduce launch for all the LayerNorms of
        # the module when the queue is flushed — instead of 2 C same-address atomics per workgroup in every launch
        part = torch.empty(L.gt_layernorm_bwd_partial_rows(R), 2 * C, dtype=torch.float32, device=dev)
        _lib.check(L.gt_layernorm_bwd_partials(_lib.ptr(a), _lib.ptr(y), 0 if y is None else y.stride(0), _lib.ptr(ln.gamma), _lib.ptr(ln.beta),
                                               _lib.ptr(rc.rowmask), _lib.ptr(mean), _lib.ptr(rstd), R, C, LN_EPS,
                                               float(p_in), int(seed_in), float(p_out), int(seed_out), int(bool(relu)) | (2 if relu_in else 0),
                                               _lib.ptr(seed_word(dev)) if (p_in > 0 or p_out > 0) else None,
                                               _lib.ptr(dout_f32), _lib.ptr(dout_bf), 0 if dout_bf is None else dout_bf.stride(0),
                                               _lib.ptr(da), _lib.ptr(dy), C, _lib.ptr(part), _st(dev)), "gt_layernorm_bwd_partials")
        q.add_ln(part, dg, db)
        grads[ln.gamma] = dg
        grads[ln.beta] = db
        return da, dy
    _lib.check(L.gt_layernorm_bwd(_lib.ptr(a), _lib.ptr(y), 0 if y is None else y.stride(0), _lib.ptr(ln.gamma), _lib.ptr(ln.beta),
                                  _lib.ptr(rc.rowmask), _lib.ptr(mean), _lib.ptr(rstd), R, C, LN_EPS,
                                  float(p_in), int(seed_in), float(p_out), int(seed_out), int(bool(relu)) | (2 if relu_in else 0),
                                  _lib.ptr(seed_word(dev)) if (p_in > 0 or p_out > 0) else None,
                                  _lib.ptr(dout_f32), _lib.ptr(dout_bf), 0 if dout_bf is None else dout_bf.stride(0),
                                  _lib.ptr(da), _lib.ptr(dy), C, _lib.ptr(dg), _lib.ptr(db), _st(dev)), "gt_layernorm_bwd")
    grads[ln.gamma] = dg
    grads[ln.beta] = db
    return da, dy


# ----------------------------------------------------------------------------- attention
def mha_fwd(rc, att, xb, p, seed):
    L = _lib.lib()
    dev = xb.device
    R = rc.R
    H, D, C = att.n_heads, att.k_channels, att.channels
    qkv = conv_rows(xb, att.pc_qkv, rc, bias=att.qkv_bias)           # one GEMM: [R, C] x [C, 3C]
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    # the attention kernel writes frame rows only: halo rows must be finite zeros (they meet zero
    # gradients in the wgrad GEMM, and 0 * NaN garbage would poison it)
    o = ops_zeros_big((R, C), torch.bfloat16, dev)
    P = torch.empty(rc.B, H, rc.T, rc.T, dtype=torch.float32, device=dev)
    Ek = att.emb_rel_k.detach().reshape(-1, D).contiguous()
    Ev = att.emb_rel_v.detach().reshape(-1, D).contiguous()
    _lib.check(L.gt_attn_fwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), 3 * C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths),
                             _lib.ptr(o), C, _lib.ptr(P), rc.B, rc.T, rc.Tp, _lib.ptr(rc.row0), H, D, att.window_size, float(p), int(seed),
                             _lib.ptr(seed_word(dev)) if p > 0 else None, _st(dev)),
               "gt_attn_fwd")
    y = conv_rows(o, att.conv_o.pc, rc, bias=att.conv_o.bias)
    return y, (xb, q, k, v, o, P, Ek, Ev, p, seed)


def mha_bwd(rc, att, saved, dy, grads):
    """dy: bf16 rows gradient of the attention block output.  Returns dxb (bf16)."""
    L = _lib.lib()
    xb, q, k, v, o, P, Ek, Ev, p, seed = saved
    dev = xb.device
    R = rc.R
    H, D, C = att.n_heads, att.k_channels, att.channels
    grads.update(conv_param_grads(att.conv_o, o, dy, R))
    do = conv_rows(dy, att.conv_o.pc, rc, dgrad=True)
    # dq | dk | dv side by side (one K = 3C data-gradient GEMM below); halo / padded rows are never written by the
    # kernel: zero the buffer once
    dqkv = ops_zeros_big((R, 3 * C), torch.bfloat16, dev)
    dq, dk, dv = dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]
    from .flow_impl import _scratch
    ws_bytes = L.gt_attn_bwd_workspace_bytes(rc.B, rc.T, H)
    ws = _scratch("attn_bwd", ws_bytes, dev)
    dEk = grad_accumulator(att.emb_rel_k, Ek.shape)
    dEv = grad_accumulator(att.emb_rel_v, Ev.shape)
    _lib.check(L.gt_attn_bwd(_lib.ptr(q), _lib.ptr(k), _lib.ptr(v), 3 * C, _lib.ptr(Ek), _lib.ptr(Ev), _lib.ptr(rc.lengths),
                             _lib.ptr(do), C, _lib.ptr(P), _lib.ptr(ws), ws_bytes, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), 3 * C,
                             _lib.ptr(dEk), _lib.ptr(dEv), rc.B, rc.T, rc.Tp, _lib.ptr(rc.row0), H, D, att.window_size, float(p), int(seed),
                             _lib.ptr(seed_word(dev)) if p > 0 else None, _st(dev)),
               "gt_attn_bwd")
    grads[att.emb_rel_k] = dEk.view_as(att.emb_rel_k)
    grads[att.emb_rel_v] = dEv.view_as(att.emb_rel_v)
    grads.update(conv_param_grads(att.conv_q, xb, dq, R))
    grads.update(conv_param_grads(att.conv_k, xb, dk, R))
    grads.update(conv_param_grads(att.conv_v, xb, dv, R))
    return conv_rows(dqkv, att.pc_qkv, rc, dgrad=True)              # [dq | dk | dv] @ [Wq; Wk; Wv]


# ----------------------------------------------------------------------------- one encoder layer
def layer_fwd(rc, enc, i, x, xb, train, seed):
    p = enc.p_dropout if train else 0.0
    att, ffn = enc.attn_layers[i], enc.ffn_layers[i]
    C = enc.hidden_channels
    y, s_att = mha_fwd(rc, att, xb, p, seed)
    x1, xb1, s_ln1 = _ln_fwd(rc, enc.norm_layers_1[i], x, y, p, seed + 1, 0.0, 0, 0, True, C)
    f1 = conv_rows(xb1, ffn.conv_1.pc, rc, bias=ffn.conv_1.bias, relu=True, mask=True, drop_p=p, seed=seed + 2)
    f2 = conv_rows(f1, ffn.conv_2.pc, rc, bias=ffn.conv_2.bias, mask=True)
    x2, xb2, s_ln2 = _ln_fwd(rc, enc.norm_layers_2[i], x1, f2, p, seed + 3, 0.0, 0, 0, True, C)
    return x2, xb2, (s_att, s_ln1, xb1, f1, s_ln2, p)


def layer_bwd(rc, enc, i, saved, dx, dxb, grads):
    """(dx fp32, dxb bf16): gradients wrt the layer's fp32 output and its bf16 copy (either may be
    None).  Returns the same pair for the layer input."""
    L = _lib.lib()
    s_att, s_ln1, xb1, f1, s_ln2, p = saved
    att, ffn = enc.attn_layers[i], enc.ffn_layers[i]
    R = rc.R
    dev = rc.device
    dx1, df2 = _ln_bwd(rc, enc.norm_layers_2[i], s_ln2, dx, dxb, True, True, grads)
    grads.update(conv_param_grads(ffn.conv_2, f1, df2, R))
    # d conv_2's input, then the backward of dropout(relu(.)) from the saved activation — in the GEMM's epilogue
    dc1 = conv_rows(df2, ffn.conv_2.pc, rc, dgrad=True, gate=3, gate_t=f1, drop_p=float(p))
    grads.update(conv_param_grads(ffn.conv_1, xb1, dc1, R))
    dxb1 = conv_rows(dc1, ffn.conv_1.pc, rc, dgrad=True)
    dx0, dy = _ln_bwd(rc, enc.norm_layers_1[i], s_ln1, dx1, dxb1, True, True, grads)
    dxb0 = mha_bwd(rc, att, s_att, dy, grads)
    return dx0, dxb0


# ----------------------------------------------------------------------------- prenet (ConvReluNorm)
def crn_fwd(rc, crn, x0, xb0, train, seed):
    L = _lib.lib()
    p = crn.p_dropout if train else 0.0
    C = crn.hidden_channels
    h = xb0
    saved = []
    for i in range(crn.n_layers):
        c = conv_rows(h, crn.conv_layers[i].pc, rc, bias=crn.conv_layers[i].bias)
        _, hn, s_ln = _ln_fwd(rc, crn.norm_layers[i], None, c, 0.0, 0, p, seed + i, 1, False, C)
        saved.append((h, s_ln))
        h = hn
    x1 = conv_rows(h, crn.proj.pc, rc, bias=crn.proj.bias, addend=x0, mask=True, out_f32=True)
    xb1 = torch.empty(rc.R, crn.out_channels, dtype=torch.bfloat16, device=rc.device)
    _lib.check(L.gt_rows_f32_to_bf16(_lib.ptr(x1), x1.stride(0), _lib.ptr(xb1), xb1.stride(0), None, rc.R, crn.out_channels,
                                     _st(rc.device)), "gt_rows_f32_to_bf16")
    return x1, xb1, (saved, h)


def _sum_grads_to_bf16(rc, dx, dxb, C, masked=True):
    """bf16( (dx + dxb) * mask ) — where the fp32 stream and its bf16 copy meet again."""
    L = _lib.lib()
    dev = rc.device
    if dx is None:
        tot = dxb.float()
    else:
        tot = dx.clone() if dxb is not None else dx
        if dxb is not None:
            _lib.check(L.gt_rows_add_bf16(_lib.ptr(tot), tot.stride(0), _lib.ptr(dxb), dxb.stride(0), rc.R, C, _st(dev)), "gt_rows_add_bf16")
    out = torch.empty(rc.R, C, dtype=torch.bfloat16, device=dev)
    _lib.check(L.gt_rows_f32_to_bf16(_lib.ptr(tot), tot.stride(0), _lib.ptr(out), C, _lib.ptr(rc.rowmask) if masked else None,
                                     rc.R, C, _st(dev)), "gt_rows_f32_to_bf16")
    return tot, out


def crn_bwd(rc, crn, saved_all, dx1, dxb1, grads):
    """Returns (dx0 fp32, dxb0 bf16)."""
    saved, hlast = saved_all
    R = rc.R
    C = crn.out_channels
    tot, dpre = _sum_grads_to_bf16(rc, dx1, dxb1, C)             # x1 = (x0 + proj(h)) * mask
    L = _lib.lib()
    dx0 = torch.empty(R, C, dtype=torch.float32, device=rc.device)
    dx0.copy_(dpre)                                               # masked sum, residual path
    grads.update(conv_param_grads(crn.proj, hlast, dpre, R))
    dh = conv_rows(dpre, crn.proj.pc, rc, dgrad=True)
    for i in reversed(range(crn.n_layers)):
        hin, s_ln = saved[i]
        _, dc = _ln_bwd(rc, crn.norm_layers[i], s_ln, None, dh, False, True, grads)
        grads.update(conv_param_grads(crn.conv_layers[i], hin, dc, R))
        dh = conv_rows(dc, crn.conv_layers[i].pc, rc, dgrad=True)
    return dx0, dh


# ----------------------------------------------------------------------------- duration predictor
def dp_fwd(rc, dp, xb, train, seed):
    p = dp.p_dropout if train else 0.0
    F = dp.filter_channels
    c1 = conv_rows(xb, dp.conv_1.pc, rc, bias=dp.conv_1.bias, relu=True)
    _, h1, s1 = _ln_fwd(rc, dp.norm_1, None, c1, 0.0, 0, p, seed, 0, False, F)
    c2 = conv_rows(h1, dp.conv_2.pc, rc, bias=dp.conv_2.bias, relu=True)
    _, h2, s2 = _ln_fwd(rc, dp.norm_2, None, c2, 0.0, 0, p, seed + 1, 0, False, F)
    out = conv_rows(h2, dp.proj_pad.pc, rc, bias=dp.proj_pad.bias, mask=True, out_f32=True)   # [R, 8], column 0 = logw
    return out, (xb, c1, s1, h1, c2, s2, h2)


def dp_bwd(rc, dp, saved, dout, grads, want_dx=False):
    """dout: [R, 8] fp32 (only column 0 non-zero).  x is detached in the reference: no input grad, except for the
    speaker vector added to it (want_dx: returns the bf16 input gradient its cond conv needs)."""
    L = _lib.lib()
    xb, c1, s1, h1, c2, s2, h2 = saved
    R = rc.R
    dev = rc.device
    F = dp.filter_channels
    db = torch.empty(R, 8, dtype=torch.bfloat16, device=dev)
    _lib.check(L.gt_rows_f32_to_bf16(_lib.ptr(dout), dout.stride(0), _lib.ptr(db), 8, _lib.ptr(rc.rowmask), R, 8, _st(dev)),
               "gt_rows_f32_to_bf16")
    g = conv_param_grads(dp.proj_pad, h2, db, R)
    grads[dp.proj.weight] = g[dp.proj_pad.weight][:1].contiguous()
    grads[dp.proj.bias] = g[dp.proj_pad.bias][:1].contiguous()
    dh2 = conv_rows(db, dp.proj_pad.pc, rc, dgrad=True)
    _, dr2 = _ln_bwd(rc, dp.norm_2, s2, None, dh2, False, True, grads, relu_in=True)      # through the norm AND the ReLU before it
    grads.update(conv_param_grads(dp.conv_2, h1, dr2, R))
    dh1 = conv_rows(dr2, dp.conv_2.pc, rc, dgrad=True)
    _, dr1 = _ln_bwd(rc, dp.norm_1, s1, None, dh1, False, True, grads, relu_in=True)
    grads.update(conv_param_grads(dp.conv_1, xb, dr1, R))
    return conv_rows(dr1, dp.conv_1.pc, rc, dgrad=True) if want_dx else None
