"""TEST INFRASTRUCTURE ONLY — CPU (PyTorch fp32/fp64) restatement of the reference's flow decoder,
text encoder and training glue, as plain functions over a parameter dict keyed by the
reference's state_dict names.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the
product package glow-tts_amd/ never does.

Parity status: PINNED to the reference's own modules — tests/golden/float_golden.npz holds
inputs/outputs/gradients produced by importing /root/reference (modules.py, attentions.py,
models.py) with closed-form weights (tests/golden/fill.py, tests/golden/make_float_golden.py);
tests/test_float_oracle.py checks every function here against it (fp32, rtol 1e-5).
The third-party arithmetic underneath (conv/matmul/softmax of PyTorch, requirements.txt pins
torch==2.0.0; this container has 2.10) is not pinned by any reference test: "parity unpinned"
at that boundary, pinned by us to container torch-CPU fp32.

Each function cites the reference lines it restates (paths relative to the reference repo).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
def sequence_mask(lengths, max_len):
    """commons.py:79-83"""
    return (torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None])


def weight_norm_w(v, g):
    """torch.nn.utils.weight_norm (dim=0) as used at modules.py:127,132,141: w = g * v / ||v||,
    norm over every dim but 0."""
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def conv_w(P, name):
    """weight of a conv that may be weight-normed (…weight_v/…weight_g) or plain (…weight)."""
    if name + ".weight_v" in P:
        return weight_norm_w(P[name + ".weight_v"], P[name + ".weight_g"])
    return P[name + ".weight"]


def conv1d(P, name, x, padding=0, dilation=1):
    return F.conv1d(x, conv_w(P, name), P.get(name + ".bias"), padding=padding, dilation=dilation)


def layer_norm_c(x, gamma, beta, eps=1e-4):
    """modules.LayerNorm (modules.py:26-44): normalise over the channel dim (1)."""
    mean = x.mean(1, keepdim=True)
    var = ((x - mean) ** 2).mean(1, keepdim=True)
    x = (x - mean) * torch.rsqrt(var + eps)
    return x * gamma.view(1, -1, 1) + beta.view(1, -1, 1)


def squeeze(x, x_mask, n_sqz=2):
    """commons.py:339-351: channel p*C + c at squeezed time t' holds x[c, n_sqz*t' + p]."""
    b, c, t = x.shape
    t = (t // n_sqz) * n_sqz
    x = x[:, :, :t]
    xs = x.view(b, c, t // n_sqz, n_sqz).permute(0, 3, 1, 2).contiguous().view(b, c * n_sqz, t // n_sqz)
    m = x_mask[:, :, n_sqz - 1::n_sqz]
    return xs * m, m


def unsqueeze(x, x_mask, n_sqz=2):
    """commons.py:354-364"""
    b, c, t = x.shape
    xu = x.view(b, n_sqz, c // n_sqz, t).permute(0, 2, 3, 1).contiguous().view(b, c // n_sqz, t * n_sqz)
    m = x_mask.unsqueeze(-1).repeat(1, 1, 1, n_sqz).view(b, 1, t * n_sqz)
    return xu * m, m


# ----------------------------------------------------------------------------- flow decoder
def actnorm_fwd(P, pre, x, x_mask):
    """modules.ActNorm.forward (modules.py:584-599), initialised (no DDI)."""
    logs, bias = P[pre + "logs"], P[pre + "bias"]
    x_len = x_mask.sum([1, 2])
    z = (bias + torch.exp(logs) * x) * x_mask
    return z, logs.sum() * x_len


def actnorm_initialize(x, x_mask):
    """modules.ActNorm.initialize (modules.py:607-619): data-dependent init from the masked batch statistics of the
    layer's input -> (logs [1,C,1], bias [1,C,1])."""
    denom = torch.sum(x_mask, [0, 2])
    m = torch.sum(x * x_mask, [0, 2]) / denom
    m_sq = torch.sum(x * x * x_mask, [0, 2]) / denom
    v = m_sq - (m ** 2)
    logs = 0.5 * torch.log(torch.clamp_min(v, 1e-6))
    return (-logs).view(1, -1, 1), (-m * torch.exp(-logs)).view(1, -1, 1)


def decoder_ddi(P, pre, x, x_mask, g=None, n_blocks=12, **kw):
    """models.FlowSpecDecoder.forward on a decoder whose ActNorms were set_ddi(True) (the reference's init.py flow): every
    ActNorm initialises from the input it sees (modules.py:588-590), block after block.  Returns a copy of P with the
    initialised logs / bias, and the forward's (z, logdet) under them."""
    P = dict(P)
    n_sqz = kw.get("n_sqz", 2)
    with torch.no_grad():
        h, m = squeeze(x, x_mask, n_sqz)
        for b in range(n_blocks):
            P[pre + f"flows.{3 * b}.logs"], P[pre + f"flows.{3 * b}.bias"] = actnorm_initialize(h, m)
            h, _ = actnorm_fwd(P, pre + f"flows.{3 * b}.", h, m)
            h, _ = invconv_fwd(P, pre + f"flows.{3 * b + 1}.", h, m, kw.get("n_split", 4))
            h, _ = coupling_fwd(P, pre + f"flows.{3 * b + 2}.", h, m, g, kw.get("n_layers", 4), kw.get("hidden", 192),
                                kw.get("kernel_size", 5), kw.get("sigmoid_scale", False))
    z, ld = decoder_fwd(P, pre, x, x_mask, g, n_blocks=n_blocks, **kw)
    return P, z, ld


def invconv_fwd(P, pre, x, x_mask, n_split=4):
    """modules.InvConvNear.forward (modules.py:635-665), restated as the grouped 4x4 mix it is
    (SURVEY App. A (ii)): for g < c/4, idx = {2g, 2g+1, c/2+2g, c/2+2g+1}: z[idx] = W @ x[idx]."""
    W = P[pre + "weight"]
    b, c, t = x.shape
    x_len = x_mask.sum([1, 2])
    h = n_split // 2
    xg = x.view(b, 2, c // n_split, h, t).permute(0, 1, 3, 2, 4).reshape(b, n_split, c // n_split, t)
    zg = torch.einsum("oi,bigt->bogt", W, xg)
    z = zg.view(b, 2, h, c // n_split, t).permute(0, 1, 3, 2, 4).reshape(b, c, t) * x_mask
    logdet = torch.logdet(W) * (c / n_split) * x_len
    return z, logdet


def gate(a, b, n):
    """commons.fused_add_tanh_sigmoid_multiply (commons.py:61-68)"""
    s = a + b
    return torch.tanh(s[:, :n]) * torch.sigmoid(s[:, n:])


def wn_fwd(P, pre, x, x_mask, g=None, n_layers=4, hidden=192, kernel_size=5, dilation_rate=1):
    """modules.WN.forward (modules.py:144-171), eval mode (dropout off)."""
    output = torch.zeros_like(x)
    if g is not None:
        g = conv1d(P, pre + "cond_layer", g)
    for i in range(n_layers):
        d = dilation_rate ** i
        pad = int((kernel_size * d - d) / 2)
        x_in = conv1d(P, pre + f"in_layers.{i}", x, padding=pad, dilation=d)
        g_l = g[:, i * 2 * hidden:(i + 1) * 2 * hidden] if g is not None else torch.zeros_like(x_in)
        acts = gate(x_in, g_l, hidden)
        rs = conv1d(P, pre + f"res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def wnp_fwd(P, pre, x, x_mask, g1=None, n_layers=4, hidden=192, kernel_size=5, dilation_rate=1, n_sqz=2):
    """modules.WNP.forward (modules.py:316-343) + WNP.squeeze (modules.py:353-362), eval mode: the identity when the
    contour g1 [b,1,t_unsqueezed] is None; else WN's loop with per-frame conditioning cond_layer1(g1), squeezed."""
    if g1 is None:
        return x
    g = conv1d(P, pre + "cond_layer1", g1)
    b, c, t = g.shape
    t = (t // n_sqz) * n_sqz
    g = g[:, :, :t].view(b, c, t // n_sqz, n_sqz).permute(0, 3, 1, 2).contiguous().view(b, c * n_sqz, t // n_sqz)
    output = torch.zeros_like(x)
    for i in range(n_layers):
        d = dilation_rate ** i
        pad = int((kernel_size * d - d) / 2)
        x_in = conv1d(P, pre + f"in_layers.{i}", x, padding=pad, dilation=d)
        acts = gate(x_in, g[:, i * 2 * hidden:(i + 1) * 2 * hidden], hidden)
        rs = conv1d(P, pre + f"res_skip_layers.{i}", acts)
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def _coupling_net(P, pre, x0, x_mask, g, pitch, energy, n_layers, hidden, kernel_size):
    """start -> wn -> wn_energy -> wn_pitch -> end (attentions.py:144-155)"""
    h = conv1d(P, pre + "start", x0) * x_mask
    h = wn_fwd(P, pre + "wn.", h, x_mask, g, n_layers, hidden, kernel_size)
    h = wnp_fwd(P, pre + "wn_energy.", h, x_mask, energy, n_layers, hidden, kernel_size)
    h = wnp_fwd(P, pre + "wn_pitch.", h, x_mask, pitch, n_layers, hidden, kernel_size)
    return conv1d(P, pre + "end", h)


def coupling_fwd(P, pre, x, x_mask, g=None, n_layers=4, hidden=192, kernel_size=5, sigmoid_scale=False,
                 pitch=None, energy=None):
    """attentions.CouplingBlock.forward (attentions.py:132-186); with pitch=energy=None
    wn_energy / wn_pitch return their input (modules.WNP.forward, modules.py:323-324)."""
    c = x.shape[1]
    x0, x1 = x[:, :c // 2], x[:, c // 2:]
    out = _coupling_net(P, pre, x0, x_mask, g, pitch, energy, n_layers, hidden, kernel_size)
    m, logs = out[:, :c // 2], out[:, c // 2:]
    if sigmoid_scale:
        logs = torch.log(1e-6 + torch.sigmoid(logs + 2))
    z1 = (m + torch.exp(logs) * x1) * x_mask
    logdet = (logs * x_mask).sum([1, 2])
    return torch.cat([x0, z1], 1), logdet


def decoder_fwd(P, pre, x, x_mask, g=None, n_blocks=12, n_layers=4, hidden=192, kernel_size=5,
                n_split=4, n_sqz=2, sigmoid_scale=False, pitch=None, energy=None):
    """models.FlowSpecDecoder.forward (models.py:765-785), reverse=False.  pitch / energy: [b,1,t] contours at the
    un-squeezed frame rate (cfg 5) or None."""
    x, m = squeeze(x, x_mask, n_sqz)
    logdet_tot = 0
    for b in range(n_blocks):
        x, ld = actnorm_fwd(P, pre + f"flows.{3 * b}.", x, m); logdet_tot = logdet_tot + ld
        x, ld = invconv_fwd(P, pre + f"flows.{3 * b + 1}.", x, m, n_split); logdet_tot = logdet_tot + ld
        x, ld = coupling_fwd(P, pre + f"flows.{3 * b + 2}.", x, m, g, n_layers, hidden, kernel_size, sigmoid_scale,
                             pitch, energy)
        logdet_tot = logdet_tot + ld
    x, _ = unsqueeze(x, m, n_sqz)
    return x, logdet_tot


# ----------------------------------------------------------------------------- reverse flow (inference)
def actnorm_rev(P, pre, x, x_mask):
    """modules.ActNorm.forward with reverse=True (modules.py:592-594)."""
    return (x - P[pre + "bias"]) * torch.exp(-P[pre + "logs"]) * x_mask


def invconv_rev(P, pre, x, x_mask, n_split=4):
    """modules.InvConvNear.forward with reverse=True (modules.py:647-652,658-664): the grouped 4x4 mix with W^-1."""
    W = torch.inverse(P[pre + "weight"].float())
    b, c, t = x.shape
    h = n_split // 2
    xg = x.view(b, 2, c // n_split, h, t).permute(0, 1, 3, 2, 4).reshape(b, n_split, c // n_split, t)
    zg = torch.einsum("oi,bigt->bogt", W, xg)
    return zg.view(b, 2, h, c // n_split, t).permute(0, 1, 3, 2, 4).reshape(b, c, t) * x_mask


def coupling_rev(P, pre, x, x_mask, g=None, n_layers=4, hidden=192, kernel_size=5, sigmoid_scale=False,
                 pitch=None, energy=None):
    """attentions.CouplingBlock.forward with reverse=True (attentions.py:178-180)."""
    c = x.shape[1]
    x0, x1 = x[:, :c // 2], x[:, c // 2:]
    out = _coupling_net(P, pre, x0, x_mask, g, pitch, energy, n_layers, hidden, kernel_size)
    m, logs = out[:, :c // 2], out[:, c // 2:]
    if sigmoid_scale:
        logs = torch.log(1e-6 + torch.sigmoid(logs + 2))
    z1 = (x1 - m) * torch.exp(-logs) * x_mask
    return torch.cat([x0, z1], 1)


def decoder_rev(P, pre, z, z_mask, g=None, n_blocks=12, n_layers=4, hidden=192, kernel_size=5,
                n_split=4, n_sqz=2, sigmoid_scale=False, pitch=None, energy=None):
    """models.FlowSpecDecoder.forward with reverse=True (models.py:765-785): flows in reverse order, no log-det."""
    x, m = squeeze(z, z_mask, n_sqz)
    for b in reversed(range(n_blocks)):
        x = coupling_rev(P, pre + f"flows.{3 * b + 2}.", x, m, g, n_layers, hidden, kernel_size, sigmoid_scale, pitch, energy)
        x = invconv_rev(P, pre + f"flows.{3 * b + 1}.", x, m, n_split)
        x = actnorm_rev(P, pre + f"flows.{3 * b}.", x, m)
    x, _ = unsqueeze(x, m, n_sqz)
    return x


def generate_path(duration, mask):
    """commons.generate_path (commons.py:127-143): token i owns frames [cum_i - d_i, cum_i)."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1)
    j = torch.arange(t_y, dtype=duration.dtype)[None, None, :]
    path = ((j < cum[:, :, None]) & (j >= (cum - duration)[:, :, None])).to(mask.dtype)
    return path * mask


# ----------------------------------------------------------------------------- text encoder
def mha_fwd(P, pre, x, c, attn_mask, n_heads=2, window_size=4):
    """attentions.MultiHeadAttention.forward/attention (attentions.py:231-272), eval mode, restated
    with the 9-diagonal band the pad/reshape skew of :292-336 amounts to:
    scores[i,j] += q_i . E_k[j-i+w] / sqrt(d),  out_i += sum_j p[i,j] E_v[j-i+w]  for |j-i| <= w."""
    q = conv1d(P, pre + "conv_q", x)
    k = conv1d(P, pre + "conv_k", c)
    v = conv1d(P, pre + "conv_v", c)
    b, d, t = q.shape
    dk = d // n_heads
    q = q.view(b, n_heads, dk, t).transpose(2, 3)
    k = k.view(b, n_heads, dk, t).transpose(2, 3)
    v = v.view(b, n_heads, dk, t).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    rel = None
    if window_size is not None:
        Ek, Ev = P[pre + "emb_rel_k"][0], P[pre + "emb_rel_v"][0]           # [2w+1, dk], shared by heads
        idx = torch.arange(t)
        rel = idx[None, :] - idx[:, None] + window_size                      # j - i + w
        band = (rel >= 0) & (rel <= 2 * window_size)
        relc = rel.clamp(0, 2 * window_size)
        qe = torch.matmul(q, Ek.t())                                         # [b,h,t,2w+1]
        bias = torch.gather(qe, 3, relc[None, None].expand(b, n_heads, t, t)) * band
        scores = scores + bias / math.sqrt(dk)
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)
    if window_size is not None:
        pw = torch.zeros(b, n_heads, t, 2 * window_size + 1, dtype=p.dtype)
        pw.scatter_add_(3, relc[None, None].expand(b, n_heads, t, t), p * band)
        out = out + torch.matmul(pw, Ev)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return conv1d(P, pre + "conv_o", out), p


def ffn_fwd(P, pre, x, x_mask, kernel_size=3):
    """attentions.FFN.forward (attentions.py:364-372), relu, eval mode."""
    x = conv1d(P, pre + "conv_1", x * x_mask, padding=kernel_size // 2)
    x = torch.relu(x)
    x = conv1d(P, pre + "conv_2", x * x_mask, padding=kernel_size // 2)
    return x * x_mask


def encoder_fwd(P, pre, x, x_mask, g=None, n_layers=6, n_heads=2, window_size=4, kernel_size=3):
    """attentions.Encoder.forward (attentions.py:56-86), eval mode."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        if i == 3 - 1 and g is not None:
            x = x + F.linear(g.transpose(2, 1), P[pre + "cond_g.weight"], P[pre + "cond_g.bias"]).transpose(2, 1)
        y, _ = mha_fwd(P, pre + f"attn_layers.{i}.", x, x, attn_mask, n_heads, window_size)
        x = layer_norm_c(x + y, P[pre + f"norm_layers_1.{i}.gamma"], P[pre + f"norm_layers_1.{i}.beta"])
        y = ffn_fwd(P, pre + f"ffn_layers.{i}.", x, x_mask, kernel_size)
        x = layer_norm_c(x + y, P[pre + f"norm_layers_2.{i}.gamma"], P[pre + f"norm_layers_2.{i}.beta"])
    return x * x_mask


def conv_relu_norm_fwd(P, pre, x, x_mask, n_layers=3, kernel_size=5):
    """modules.ConvReluNorm.forward (modules.py:95-102), eval mode."""
    x_org = x
    for i in range(n_layers):
        x = conv1d(P, pre + f"conv_layers.{i}", x * x_mask, padding=kernel_size // 2)
        x = layer_norm_c(x, P[pre + f"norm_layers.{i}.gamma"], P[pre + f"norm_layers.{i}.beta"])
        x = torch.relu(x)
    x = x_org + conv1d(P, pre + "proj", x)
    return x * x_mask


def text_encoder_fwd(P, pre, ids, x_lengths, g=None, hidden=192, n_layers=6, n_heads=2, window_size=4,
                     kernel_size=3, prenet=True, mean_only=True, l=None):
    """models.TextEncoder.forward (models.py:692-716), eval mode.  l [b, lin, 1]: language vector concatenated to every
    position of the (lin channels narrower) token embedding (models.py:698-699)."""
    x = F.embedding(ids, P[pre + "emb.weight"]) * math.sqrt(hidden)
    if l is not None:
        x = torch.cat((x, l.transpose(2, 1).expand(x.size(0), x.size(1), -1)), dim=-1)
    x = x.transpose(1, -1)
    x_mask = sequence_mask(x_lengths, x.size(2)).unsqueeze(1).to(x.dtype)
    if prenet:
        x = conv_relu_norm_fwd(P, pre + "pre.", x, x_mask)
    x = encoder_fwd(P, pre + "encoder.", x, x_mask, g, n_layers, n_heads, window_size, kernel_size)
    x_m = conv1d(P, pre + "proj_m", x) * x_mask
    x_logs = conv1d(P, pre + "proj_s", x) * x_mask if not mean_only else torch.zeros_like(x_m)
    return x, x_m, x_logs, x_mask


def duration_predictor_fwd(P, pre, x, x_mask, kernel_size=3, g=None, l=None):
    """models.DurationPredictor.forward (models.py:585-612), l=emo=None, eval mode; g [b,gin,1] is detached and
    added through the 1x1 `cond` conv (models.py:587-589)."""
    x = x.detach()
    if g is not None:
        x = x + conv1d(P, pre + "cond", g.detach())
    if l is not None:                                     # models.py:595-597
        x = x + conv1d(P, pre + "cond_lang", l.detach())
    x = conv1d(P, pre + "conv_1", x * x_mask, padding=kernel_size // 2)
    x = layer_norm_c(torch.relu(x), P[pre + "norm_1.gamma"], P[pre + "norm_1.beta"])
    x = conv1d(P, pre + "conv_2", x * x_mask, padding=kernel_size // 2)
    x = layer_norm_c(torch.relu(x), P[pre + "norm_2.gamma"], P[pre + "norm_2.beta"])
    x = conv1d(P, pre + "proj", x * x_mask)
    return x * x_mask


# ----------------------------------------------------------------------------- stochastic predictors
def layer_norm2(x, gamma, beta, eps=1e-5):
    """modules.LayerNorm2 (modules.py:46-68): F.layer_norm over the channel dim, eps 1e-5."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def dds_conv(P, pre, x, x_mask, g=None, kernel_size=3, num_layers=3):
    """modules.DilatedDepthSeparableConv.forward (modules.py:718-735), eval mode (dropout off)."""
    if g is not None:
        x = x + g
    for i in range(num_layers):
        d = kernel_size ** i
        pad = (kernel_size * d - d) // 2
        y = F.conv1d(x * x_mask, P[pre + f"convs_sep.{i}.weight"], P[pre + f"convs_sep.{i}.bias"], padding=pad, dilation=d,
                     groups=x.shape[1])
        y = F.gelu(layer_norm2(y, P[pre + f"norms_1.{i}.gamma"], P[pre + f"norms_1.{i}.beta"]))
        y = F.conv1d(y, P[pre + f"convs_1x1.{i}.weight"], P[pre + f"convs_1x1.{i}.bias"])
        y = F.gelu(layer_norm2(y, P[pre + f"norms_2.{i}.gamma"], P[pre + f"norms_2.{i}.beta"]))
        x = x + y
    return x * x_mask


def rq_spline_fwd(inputs, uw, uh, ud, tail_bound=5.0, min_bin=1e-3, min_der=1e-3):
    """transforms.piecewise_rational_quadratic_transform(inverse=False, tails="linear") (transforms.py:12-202): inputs [...],
    uw / uh [..., K], ud [..., K-1] -> (outputs, logabsdet).  Restated without boolean-mask scatter: the spline is
    evaluated everywhere on clamped inputs and the linear tails are selected afterwards (same values inside and outside)."""
    K = uw.shape[-1]
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    const = math.log(math.exp(1 - min_der) - 1)
    ud = F.pad(ud, (1, 1), value=const)
    left = bottom = -tail_bound
    right = top = tail_bound
    x = inputs.clamp(left, right)
    widths = min_bin + (1 - min_bin * K) * F.softmax(uw, dim=-1)
    cumw = F.pad(torch.cumsum(widths, -1), (1, 0)) * (right - left) + left
    cumw = torch.cat([torch.full_like(cumw[..., :1], left), cumw[..., 1:-1], torch.full_like(cumw[..., :1], right)], -1)
    widths = cumw[..., 1:] - cumw[..., :-1]
    der = min_der + F.softplus(ud)
    heights = min_bin + (1 - min_bin * K) * F.softmax(uh, dim=-1)
    cumh = F.pad(torch.cumsum(heights, -1), (1, 0)) * (top - bottom) + bottom
    cumh = torch.cat([torch.full_like(cumh[..., :1], bottom), cumh[..., 1:-1], torch.full_like(cumh[..., :1], top)], -1)
    heights = cumh[..., 1:] - cumh[..., :-1]
    loc = cumw.detach().clone(); loc[..., -1] += 1e-6                          # transforms.searchsorted (:46-48)
    idx = (torch.sum(x[..., None] >= loc, -1) - 1).clamp(0, K - 1)[..., None]
    in_cw, in_w = cumw.gather(-1, idx)[..., 0], widths.gather(-1, idx)[..., 0]
    in_ch, in_h = cumh.gather(-1, idx)[..., 0], heights.gather(-1, idx)[..., 0]
    delta = (heights / widths).gather(-1, idx)[..., 0]
    d0, d1 = der.gather(-1, idx)[..., 0], der[..., 1:].gather(-1, idx)[..., 0]
    theta = (x - in_cw) / in_w
    t1 = theta * (1 - theta)
    num = in_h * (delta * theta.pow(2) + d0 * t1)
    den = delta + (d0 + d1 - 2 * delta) * t1
    out = in_ch + num / den
    dnum = delta.pow(2) * (d1 * theta.pow(2) + 2 * delta * t1 + d0 * (1 - theta).pow(2))
    lad = torch.log(dnum) - 2 * torch.log(den)
    return torch.where(inside, out, inputs), torch.where(inside, lad, torch.zeros_like(lad))


def rq_spline_inv(inputs, uw, uh, ud, tail_bound=5.0, min_bin=1e-3, min_der=1e-3):
    """the inverse=True branch of the same transform (transforms.py:152-180) -> outputs only (synthesis)."""
    K = uw.shape[-1]
    inside = (inputs >= -tail_bound) & (inputs <= tail_bound)
    const = math.log(math.exp(1 - min_der) - 1)
    ud = F.pad(ud, (1, 1), value=const)
    lo, hi = -tail_bound, tail_bound
    y = inputs.clamp(lo, hi)
    widths = min_bin + (1 - min_bin * K) * F.softmax(uw, dim=-1)
    cumw = F.pad(torch.cumsum(widths, -1), (1, 0)) * (hi - lo) + lo
    cumw[..., 0] = lo; cumw[..., -1] = hi
    widths = cumw[..., 1:] - cumw[..., :-1]
    der = min_der + F.softplus(ud)
    heights = min_bin + (1 - min_bin * K) * F.softmax(uh, dim=-1)
    cumh = F.pad(torch.cumsum(heights, -1), (1, 0)) * (hi - lo) + lo
    cumh[..., 0] = lo; cumh[..., -1] = hi
    heights = cumh[..., 1:] - cumh[..., :-1]
    loc = cumh.clone(); loc[..., -1] += 1e-6
    idx = (torch.sum(y[..., None] >= loc, -1) - 1).clamp(0, K - 1)[..., None]
    in_cw, in_w = cumw.gather(-1, idx)[..., 0], widths.gather(-1, idx)[..., 0]
    in_ch, in_h = cumh.gather(-1, idx)[..., 0], heights.gather(-1, idx)[..., 0]
    delta = (heights / widths).gather(-1, idx)[..., 0]
    d0, d1 = der.gather(-1, idx)[..., 0], der[..., 1:].gather(-1, idx)[..., 0]
    a = (y - in_ch) * (d0 + d1 - 2 * delta) + in_h * (delta - d0)
    b = in_h * d0 - (y - in_ch) * (d0 + d1 - 2 * delta)
    c = -delta * (y - in_ch)
    root = (2 * c) / (-b - torch.sqrt(b.pow(2) - 4 * a * c))
    return torch.where(inside, root * in_w + in_cw, inputs)


def conv_flow(P, pre, x, x_mask, g, reverse=False, hidden=192, num_bins=10, tail_bound=5.0):
    """modules.ConvFlow.forward (modules.py:792-819), in_channels = 2."""
    x0, x1 = x[:, :1], x[:, 1:]
    h = F.conv1d(x0, P[pre + "pre.weight"], P[pre + "pre.bias"])
    h = dds_conv(P, pre + "convs.", h, x_mask, g=g)
    h = F.conv1d(h, P[pre + "proj.weight"], P[pre + "proj.bias"]) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    uw = h[..., :num_bins] / math.sqrt(hidden)
    uh = h[..., num_bins:2 * num_bins] / math.sqrt(hidden)
    ud = h[..., 2 * num_bins:]
    if reverse:
        x1 = rq_spline_inv(x1, uw, uh, ud, tail_bound)
        return torch.cat([x0, x1], 1) * x_mask
    x1, lad = rq_spline_fwd(x1, uw, uh, ud, tail_bound)
    return torch.cat([x0, x1], 1) * x_mask, torch.sum(lad * x_mask, [1, 2])


def elementwise_affine(P, pre, x, x_mask, reverse=False):
    """modules.ElementwiseAffine.forward (modules.py:750-756)."""
    if reverse:
        return (x - P[pre + "translation"]) * torch.exp(-P[pre + "log_scale"]) * x_mask
    y = (x * torch.exp(P[pre + "log_scale"]) + P[pre + "translation"]) * x_mask
    return y, torch.sum(P[pre + "log_scale"] * x_mask, [1, 2])


def _flow_stack(P, pre, z, x_mask, g, n_flows=4):
    """flows[0] = ElementwiseAffine, flows[1:] = ConvFlow, each ConvFlow followed by a channel flip (models.py:314-318)"""
    logdet_tot = 0
    for idx in range(n_flows + 1):
        if idx == 0:
            z, ld = elementwise_affine(P, pre + "0.", z, x_mask)
        else:
            z, ld = conv_flow(P, pre + f"{idx}.", z, x_mask, g)
            z = torch.flip(z, [1])
        logdet_tot = logdet_tot + ld
    return z, logdet_tot


def _predictor_cond(P, pre, x, x_mask, g=None, l=None):
    """pre -> (+ cond(g)) (+ cond_lang(l)) -> convs -> proj * mask (models.py:262-278, 365-377): the text-side condition"""
    x = F.conv1d(x.detach(), P[pre + "pre.weight"], P[pre + "pre.bias"])
    if g is not None:
        x = x + F.conv1d(g.detach(), P[pre + "cond.weight"], P[pre + "cond.bias"])
    if l is not None:
        x = x + F.conv1d(l.detach(), P[pre + "cond_lang.weight"], P[pre + "cond_lang.bias"])
    x = dds_conv(P, pre + "convs.", x, x_mask)
    return F.conv1d(x, P[pre + "proj.weight"], P[pre + "proj.bias"]) * x_mask


def sdp_fwd(P, pre, x, x_mask, w, noise, g=None, l=None, n_flows=4):
    """models.StochasticDurationPredictor.forward, reverse=False (models.py:261-322), eval mode; `noise` [b,2,t] replaces
    the torch.randn draw of :288 (the reference multiplies it by x_mask itself).  Returns the per-utterance nll [b]."""
    x = _predictor_cond(P, pre, x, x_mask, g, l)
    h = F.conv1d(w, P[pre + "post_pre.weight"], P[pre + "post_pre.bias"])
    h = dds_conv(P, pre + "post_convs.", h, x_mask)
    h = F.conv1d(h, P[pre + "post_proj.weight"], P[pre + "post_proj.bias"]) * x_mask
    e_q = noise * x_mask
    z_q, logdet_tot_q = _flow_stack(P, pre + "post_flows.", e_q, x_mask, x + h, n_flows)
    z_u, z_v = z_q[:, :1], z_q[:, 1:]
    u = torch.sigmoid(z_u) * x_mask
    z0 = (w - u) * x_mask
    logdet_tot_q = logdet_tot_q + torch.sum((F.logsigmoid(z_u) + F.logsigmoid(-z_u)) * x_mask, [1, 2])
    nll_post = torch.sum(-0.5 * (math.log(2 * math.pi) + (e_q ** 2)) * x_mask, [1, 2]) - logdet_tot_q
    z0 = torch.log(torch.clamp_min(z0, 1e-5)) * x_mask
    logdet_tot = torch.sum(-z0, [1, 2])
    z = torch.cat([z0, z_v], 1)
    z, ld = _flow_stack(P, pre + "flows.", z, x_mask, x, n_flows)
    logdet_tot = logdet_tot + ld
    nll = torch.sum(0.5 * (math.log(2 * math.pi) + (z ** 2)) * x_mask, [1, 2]) - logdet_tot
    return nll + nll_post


def spp_fwd(P, pre, x, x_mask, dr, noise, g=None, n_flows=4):
    """models.StochasticPitchPredictor / StochasticEnergyPredictor.forward, reverse=False (models.py:364-396, 438-470):
    z = cat(dr, noise * mask) through the flow stack conditioned on the text-side features; nll [b]."""
    x = _predictor_cond(P, pre, x, x_mask, g)
    z = torch.cat([dr, noise * x_mask], 1)
    z, logdet_tot = _flow_stack(P, pre + "flows.", z, x_mask, x, n_flows)
    return torch.sum(0.5 * (math.log(2 * math.pi) + (z ** 2)) * x_mask, [1, 2]) - logdet_tot


def predictor_reverse(P, pre, x, x_mask, noise, g=None, l=None, n_flows=4):
    """reverse=True branch shared by the three predictors (models.py:324-333, 398-407): flows reversed with the useless
    vflow dropped, z = noise (already scaled), flip before every flow; returns channel 0 = logw / log-f0 / log-energy."""
    x = _predictor_cond(P, pre, x, x_mask, g, l)
    order = list(reversed(range(n_flows + 1)))
    order = order[:-2] + [order[-1]]
    z = noise
    for idx in order:
        z = torch.flip(z, [1])
        z = elementwise_affine(P, pre + "flows.0.", z, x_mask, reverse=True) if idx == 0 else \
            conv_flow(P, pre + f"flows.{idx}.", z, x_mask, x, reverse=True)
    return z[:, :1]


# ----------------------------------------------------------------------------- cfg 5 front end
def emotion_speaker_vector(P, g, emo, emo_cartesian):
    """models.FlowGenerator.forward front end (models.py:1008-1042): g [b,512] raw speaker embedding, emo [b] int64,
    emo_cartesian [b,3] -> the conditioning vector [b, gin, 1] that encoder / predictors / decoder see."""
    g = F.linear(F.normalize(g), P["emb_g.weight"], P["emb_g.bias"])
    emos_proj = F.linear(F.embedding(emo, P["emo_id_proj.weight"]), P["emo_proj.weight"], P["emo_proj.bias"])
    intens = F.linear(emo_cartesian[:, :1], P["emo_VAD_inten_proj.weight"], P["emo_VAD_inten_proj.bias"])
    ele = F.embedding(torch.bucketize(emo_cartesian[:, 1].contiguous(), P["elevation_bins"]), P["elevation_emb.weight"])
    azi = F.embedding(torch.bucketize(emo_cartesian[:, 2].contiguous(), P["azimuth_bins"]), P["azimuth_emb.weight"])
    style = F.linear(torch.cat((ele, azi), -1), P["sty_proj.weight"], P["sty_proj.bias"])
    emosty = F.layer_norm(F.softplus(torch.cat((emos_proj, style), -1)), (style.shape[-1] * 2,),
                          P["emosty_layer_norm.weight"], P["emosty_layer_norm.bias"])
    return torch.cat((g, intens + emosty), -1).unsqueeze(-1)


# ----------------------------------------------------------------------------- training glue
def logp_lattice(x_m, x_logs, z):
    """models.py:1076-1082"""
    x_s_sq_r = torch.exp(-2 * x_logs)
    logp1 = torch.sum(-0.5 * math.log(2 * math.pi) - x_logs, [1]).unsqueeze(-1)
    logp2 = torch.matmul(x_s_sq_r.transpose(1, 2), -0.5 * (z ** 2))
    logp3 = torch.matmul((x_m * x_s_sq_r).transpose(1, 2), z)
    logp4 = torch.sum(-0.5 * (x_m ** 2) * x_s_sq_r, [1]).unsqueeze(-1)
    return logp1 + logp2 + logp3 + logp4


def mle_loss(z, m, logs, logdet, mask):
    """commons.py:28-33"""
    l = torch.sum(logs) + 0.5 * torch.sum(torch.exp(-2 * logs) * ((z - m) ** 2))
    l = l - torch.sum(logdet)
    l = l / torch.sum(torch.ones_like(z) * mask)
    return l + 0.5 * math.log(2 * math.pi)


def contour_norm(c, y_max):
    """models.py:1054-1071: raw pitch / energy [b,1,t] -> log(clamp(., tiny)) with zeros kept at 0, [b,1,y_max]."""
    if c is None:
        return None
    c = c.squeeze(1)[:, :y_max]
    zero = c == 0.0
    n = torch.log(torch.clamp(c, min=torch.finfo(c.dtype).tiny)).clone()
    n[zero] = 0.0
    return n.unsqueeze(1)


def train_forward(P, ids, x_lengths, y, y_lengths, maximum_path, hp, g=None, pitch=None, energy=None, l=None):
    """The upstream-equivalent live sub-graph of models.FlowGenerator.forward
    (models.py:1050-1119) for the base configs (the fork's FlowGenerator does not
    construct for them): TextEncoder -> FlowSpecDecoder -> logp -> MAS -> duration loss
    (deterministic DurationPredictor, models.py:1089-1092) -> prior expansion -> mle loss.
    `maximum_path(value, mask) -> path` is the MAS implementation to use (tests pass the oracle).
    g [b,gin,1]: the speaker vector of the multi-speaker configs as it reaches the encoder / duration predictor /
    decoder (models.py:1046,1075,1090).  pitch / energy: raw contours [b,1,t_y] of cfg 5 into the decoder's WNPs (their
    predictor losses, SURVEY §8 f1, are not part of this sub-graph).  l [b, lin, 1]: the language vector emb_l(lang id)
    (models.py:1011-1012) into the text encoder and the duration predictor."""
    n_sqz = hp.get("n_sqz", 2)
    x, x_m, x_logs, x_mask = text_encoder_fwd(P, "encoder.", ids, x_lengths, g, hp["hidden_channels"],
                                              hp["n_layers_enc"], hp["n_heads"], hp["window_size"],
                                              hp["kernel_size"], hp["prenet"], hp["mean_only"], l=l)
    y_max = (y.size(2) // n_sqz) * n_sqz                                     # models.py:1248-1253
    y = y[:, :, :y_max]
    y_lengths = (y_lengths // n_sqz) * n_sqz
    z_mask = sequence_mask(y_lengths, y_max).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
    z, logdet = decoder_fwd(P, "decoder.", y, z_mask, g, hp["n_blocks_dec"], hp["n_block_layers"],
                            hp["hidden_channels"], hp["kernel_size_dec"], 4, n_sqz,
                            pitch=contour_norm(pitch, y_max), energy=contour_norm(energy, y_max))
    with torch.no_grad():
        logp = logp_lattice(x_m, x_logs, z)
        attn = maximum_path(logp, attn_mask.squeeze(1)).unsqueeze(1).detach()
    w = attn.squeeze(1).sum(2).unsqueeze(1)
    logw_ = torch.log(w + 1e-8) * x_mask
    logw = duration_predictor_fwd(P, "encoder.proj_w.", x, x_mask, hp["kernel_size"], g, l)
    l_length = torch.sum((logw - logw_) ** 2, [1, 2]) / torch.sum(x_mask)
    z_m = torch.matmul(attn.squeeze(1).transpose(1, 2), x_m.transpose(1, 2)).transpose(1, 2)
    z_logs = torch.matmul(attn.squeeze(1).transpose(1, 2), x_logs.transpose(1, 2)).transpose(1, 2)
    l_mle = mle_loss(z, z_m, z_logs, logdet, z_mask)
    return dict(z=z, z_m=z_m, z_logs=z_logs, logdet=logdet, z_mask=z_mask, x_m=x_m, x_mask=x_mask,
                attn=attn, logp=logp, l_length=l_length, l_mle=l_mle, loss=l_mle + torch.sum(l_length))


def train_forward_full(P, ids, x_lengths, y, y_lengths, maximum_path, hp, g, emo, emo_cartesian, pitch, energy, lids, noises,
                       x_for_predictors=None):
    """models.FlowGenerator.forward as the fork runs it for configs/base_blank_emo_lang_pitch.json (models.py:1007-1133):
    speaker / emotion front end, language embedding, TextEncoder, FlowSpecDecoder with the pitch / energy WaveNets, logp,
    MAS, StochasticDurationPredictor loss, x_feature = x @ attn, stochastic pitch / energy predictor losses, prior expansion.
    noises = (e_w [b,2,t_x], e_p [b,1,t_y], e_e [b,1,t_y]) replace the three torch.randn draws (models.py:288,383,457).
    x_for_predictors: the text-encoder output the three predictors read (they detach it, models.py:262,365,439); a test may
    hand in the product's own (bf16-stored) encoder output so that the predictors — whose spline flows are chaotically
    sensitive to their conditioning — are compared on identical inputs.
    Returns the reference's 5-tuple entries by name plus the training loss of train_ms_emo_lang_pitch.py:295-306."""
    n_sqz = hp.get("n_sqz", 2)
    gv = emotion_speaker_vector(P, g, emo, emo_cartesian)                     # [b, gin, 1]
    lv = F.embedding(lids, P["emb_l.weight"]).unsqueeze(-1)                   # [b, lin, 1]
    x, x_m, x_logs, x_mask = text_encoder_fwd(P, "encoder.", ids, x_lengths, gv, hp["hidden_channels"], hp["n_layers_enc"],
                                              hp["n_heads"], hp["window_size"], hp["kernel_size"], hp["prenet"],
                                              hp["mean_only"], l=lv)
    y_max = (y.size(2) // n_sqz) * n_sqz
    y = y[:, :, :y_max]
    y_lengths = (y_lengths // n_sqz) * n_sqz
    z_mask = sequence_mask(y_lengths, y_max).unsqueeze(1).to(x_mask.dtype)
    attn_mask = x_mask.unsqueeze(-1) * z_mask.unsqueeze(2)
    pitch_norm, energy_norm = contour_norm(pitch, y_max), contour_norm(energy, y_max)
    z, logdet = decoder_fwd(P, "decoder.", y, z_mask, gv, hp["n_blocks_dec"], hp["n_block_layers"], hp["hidden_channels"],
                            hp["kernel_size_dec"], 4, n_sqz, pitch=pitch_norm, energy=energy_norm)
    with torch.no_grad():
        logp = logp_lattice(x_m, x_logs, z)
        attn = maximum_path(logp, attn_mask.squeeze(1)).unsqueeze(1).detach()
    w = attn.squeeze(1).sum(2).unsqueeze(1)
    xp = x if x_for_predictors is None else x_for_predictors
    l_length = sdp_fwd(P, "encoder.proj_w.", xp, x_mask, w, noises[0], g=gv, l=lv) / torch.sum(x_mask)
    x_feature = torch.matmul(xp, attn.squeeze(1))
    l_pitch = torch.sum(spp_fwd(P, "proj_pitch.", x_feature, z_mask, pitch_norm, noises[1], g=gv) / torch.sum(z_mask))
    l_energy = torch.sum(spp_fwd(P, "proj_energy.", x_feature, z_mask, energy_norm, noises[2], g=gv) / torch.sum(z_mask))
    z_m = torch.matmul(attn.squeeze(1).transpose(1, 2), x_m.transpose(1, 2)).transpose(1, 2)
    z_logs = torch.matmul(attn.squeeze(1).transpose(1, 2), x_logs.transpose(1, 2)).transpose(1, 2)
    l_mle = mle_loss(z, z_m, z_logs, logdet, z_mask)
    loss = l_mle + torch.sum(l_length) + 0.5 * l_pitch + 0.5 * l_energy
    return dict(z=z, z_m=z_m, z_logs=z_logs, logdet=logdet, z_mask=z_mask, x_m=x_m, x_mask=x_mask, attn=attn, logp=logp,
                l_length=l_length, l_pitch=l_pitch, l_energy=l_energy, l_mle=l_mle, loss=loss, g=gv, x=x, x_feature=x_feature)
