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
