"""Host-side mirror of the reference's attentions.py for the hot path (same names, constructor
arguments and state_dict keys); arithmetic in HIP kernels behind the C-ABI.

  CouplingBlock         reference attentions.py:89-194
  MultiHeadAttention    reference attentions.py:197-347
  FFN / Encoder         reference attentions.py:350-372 / 12-86
"""
import torch
from torch import nn

from . import flow_impl, wgrad
from .modules import WN, WNP, ConvP, WNConvP, _RowsFn, _mask_lengths, prepare_all
from .ops import PackedConv, PackSliceN, RowsCtx


def _wn_cond(wn, g):
    """cond_layer(g) (modules.py:148-149): a [B,gin]x[gin,2*H*n] product on B rows — host-side
    PyTorch plumbing (B <= 128 rows; differentiable w.r.t. g and the cond_layer parameters)."""
    if g is None:
        return None
    cl = wn.cond_layer
    v = cl.weight_v
    w = v * (cl.weight_g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1))
    return torch.nn.functional.linear(g.squeeze(-1), w.squeeze(-1), cl.bias)     # (a matmul: MIOpen's 1x1 conv path is slow here)


def _wn_cond_all(wns, g):
    """_wn_cond for every coupling block of the decoder at once: one batched [n, B, gin] x [n, gin, 2*H*layers]
    product instead of n small ones (same arithmetic; ~10x fewer launches in the step)."""
    if g is None:
        return [None] * len(wns)
    v = torch.stack([w.cond_layer.weight_v.squeeze(-1) for w in wns])          # [n, O, gin]
    gg = torch.stack([w.cond_layer.weight_g.reshape(-1) for w in wns])         # [n, O]
    bias = torch.stack([w.cond_layer.bias for w in wns])                       # [n, O]
    w = v * (gg / v.norm(dim=2)).unsqueeze(-1)
    out = torch.einsum("bg,nog->nbo", g.squeeze(-1), w) + bias[:, None, :]
    return list(out.unbind(0))


class CouplingBlock(nn.Module):
    """reference attentions.CouplingBlock (attentions.py:89-194).  `with_prosody_wn=True` also
    creates the fork's wn_pitch / wn_energy parameter containers (state_dict compatibility; they are
    the identity when pitch/energy are None, modules.py:323-324 — SURVEY F4)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0,
                 emoin_channels=0, p_dropout=0, sigmoid_scale=False, n_sqz=2, with_prosody_wn=False):
        super().__init__()
        self.in_channels, self.hidden_channels, self.kernel_size = in_channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.gin_channels = dilation_rate, n_layers, gin_channels
        self.p_dropout, self.sigmoid_scale = p_dropout, sigmoid_scale
        self.start = WNConvP(in_channels // 2, hidden_channels, 1)
        self.end = ConvP(hidden_channels, in_channels, 1, zero_init=True)       # attentions.py:107-109
        self.wn = WN(in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels, p_dropout)
        if with_prosody_wn:                                                      # attentions.py:113-114
            self.wn_pitch = WNP(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout, 1, n_sqz)
            self.wn_energy = WNP(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout, 1, n_sqz)

    def set_boundary_fused(self, on):
        """Keep fragment-ordered twins of the start / end / skip-cat images for the fused between-WaveNets kernels
        (csrc/wn_boundary.hip; the owning FlowSpecDecoder decides).  Returns whether the block's shape qualifies."""
        ok = bool(on) and self.in_channels == 160 and self.hidden_channels == 192 and self.n_layers == 4 and \
            getattr(self.wn, "fused", False)
        self.start.also_frag = self.end.also_frag = ok
        # whichever WaveNet runs last in the block (wn, or wn_energy / wn_pitch when their contours are given) feeds the boundary
        # kernel's skip GEMM: all of them keep the fragment-ordered twin of their skip-cat image
        for w in (self.wn, getattr(self, "wn_energy", None), getattr(self, "wn_pitch", None)):
            if w is not None:
                w.set_boundary_frag(ok and getattr(w, "fused", False))
        return ok

    def store_inverse(self):
        """attentions.py:188-194 removes the weight norms of wn / wn_energy / wn_pitch so that synthesis stops
        recomputing g*v/|v|: here that product only exists inside the packed bf16 images, which FlowSpecDecoder.store_inverse
        freezes (one pack, reused by every reverse call); the parameters keep their weight_g / weight_v form."""
        prepare_all(self)
        self._frozen = True

    def _cond_affine(self, which):
        """cond_layer1 (one input channel, weight-normed 1x1 conv: modules.py:289-291,320) as the per-frame affine map
        (w [O], b [O]); differentiable w.r.t. its parameters."""
        c = getattr(self, which).cond_layer1
        v = c.weight_v.reshape(c.out_channels)
        return torch.stack([v * (c.weight_g.reshape(-1) / v.abs()), c.bias])

    def forward(self, x, x_mask=None, reverse=False, g=None, emo=None, pitch=None, energy=None, **kwargs):
        """attentions.py:132-186 as a stand-alone module (inside models.FlowSpecDecoder the whole chain is one node):
        forward -> (z, logdet); reverse=True -> (x, None), no autograd.  pitch / energy [b,1,2t] (or [b,2t]): contours at
        the UN-squeezed frame rate for wn_pitch / wn_energy (needs with_prosody_wn=True)."""
        if (pitch is not None or energy is not None) and not hasattr(self, "wn_pitch"):
            raise ValueError("pitch / energy conditioning needs CouplingBlock(with_prosody_wn=True)")
        if x_mask is None:
            x_mask = torch.ones(x.shape[0], 1, x.shape[2], device=x.device, dtype=x.dtype)
        pitch = pitch.unsqueeze(1) if (pitch is not None and pitch.dim() == 2) else pitch
        energy = energy.unsqueeze(1) if (energy is not None and energy.dim() == 2) else energy
        if not (reverse and getattr(self, "_frozen", False)):
            prepare_all(self)
            self._frozen = False
        runner = _CouplingRunner(self, x_mask, g is not None, self.training and not reverse, energy=energy, pitch=pitch)
        affs = [self._cond_affine(w) for w, c in (("wn_energy", energy), ("wn_pitch", pitch)) if c is not None]
        tensors = [x] + ([_wn_cond(self.wn, g)] if g is not None else []) + affs + runner.params
        if reverse:
            with torch.no_grad():
                return runner.reverse(*[t.detach() for t in tensors]), None
        z, logdet = _RowsFn.apply(runner, 2, *tensors)
        return z, logdet


class _CouplingRunner:
    def __init__(self, cb, x_mask, has_cond, train, seed=0, energy=None, pitch=None):
        self.cb, self.has_cond, self.train, self.seed = cb, has_cond, train, seed
        self.x_mask, self.energy, self.pitch = x_mask, energy, pitch
        self.params = [p for n, p in cb.named_parameters() if not n.startswith("wn.cond_layer") and "cond_layer1" not in n]

    def _inputs(self, x, rest):
        k = 0
        cond = eaff = paff = None
        if self.has_cond:
            cond = rest[k]; k += 1
        if self.energy is not None:
            eaff = rest[k]; k += 1
        if self.pitch is not None:
            paff = rest[k]; k += 1
        B, C, T = x.shape
        rc = RowsCtx(_mask_lengths(self.x_mask), T)
        xr = rc.to_rows(x.detach().float() * self.x_mask)
        esig, psig = flow_impl.contour_rows(rc, self.energy, B, 2 * T), flow_impl.contour_rows(rc, self.pitch, B, 2 * T)
        return rc, xr, cond, esig, psig, flow_impl.cond_rows(esig, eaff), flow_impl.cond_rows(psig, paff)

    def forward(self, x, *rest):
        rc, xr, cond, esig, psig, econd, pcond = self._inputs(x, rest)
        x0 = xr[:, :x.shape[1] // 2].to(torch.bfloat16)
        logdet = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        z, saved = flow_impl.coupling_fwd(rc, self.cb, xr, x0, cond, logdet, self.train, self.seed, econd=econd, pcond=pcond)
        return (rc.from_rows(z), logdet), (rc, saved, esig, psig)

    def reverse(self, z, *rest):
        rc, zr, cond, _, _, econd, pcond = self._inputs(z, rest)
        z0 = zr[:, :z.shape[1] // 2].to(torch.bfloat16)
        return rc.from_rows(flow_impl.coupling_rev(rc, self.cb, zr, z0, cond, econd=econd, pcond=pcond)).to(z.dtype)

    def backward(self, saved_all, dz, dlogdet):
        rc, saved, esig, psig = saved_all
        dlogdet = torch.zeros(rc.B, device=dz.device) if dlogdet is None else dlogdet.contiguous().float()
        dzr = rc.to_rows(dz.float())
        with wgrad.WgradQueue(dz.device, site=self.cb):
            if esig is None and psig is None:
                dx, grads, dcond = flow_impl.coupling_bwd(rc, self.cb, saved, dzr, dlogdet, self.has_cond)
                dpros = [None, None]
            else:
                dx, grads, dcond, dpros = flow_impl.coupling_bwd(rc, self.cb, saved, dzr, dlogdet, self.has_cond,
                                                                 econd=esig is not None, pcond=psig is not None)
        out = [rc.from_rows(dx)]
        if self.has_cond:
            out.append(dcond)
        out += [flow_impl.cond_affine_grads(dc, sig) for dc, sig in ((dpros[0], esig), (dpros[1], psig)) if sig is not None]
        return out + [grads.get(p) for p in self.params]


# ======================================================================================= text encoder
from . import encoder_impl  # noqa: E402
from .modules import LayerNorm  # noqa: E402


class MultiHeadAttention(nn.Module):
    """reference attentions.MultiHeadAttention (attentions.py:197-347), self-attention with the
    windowed relative-position terms (heads_share=True as in every reference use)."""

    def __init__(self, channels, out_channels, n_heads, window_size=None, heads_share=True, p_dropout=0.,
                 block_length=None, proximal_bias=False, proximal_init=False):
        super().__init__()
        assert channels % n_heads == 0
        assert window_size is not None and heads_share and block_length is None and not proximal_bias, \
            "kernels implement the configuration every reference config uses (window_size=4, shared heads)"
        self.channels, self.out_channels, self.n_heads, self.window_size = channels, out_channels, n_heads, window_size
        self.heads_share, self.block_length, self.proximal_bias, self.p_dropout = heads_share, block_length, proximal_bias, p_dropout
        self.attn = None
        self.k_channels = channels // n_heads
        self.conv_q = ConvP(channels, channels, 1)
        self.conv_k = ConvP(channels, channels, 1)
        self.conv_v = ConvP(channels, channels, 1)
        rel_stddev = self.k_channels ** -0.5
        self.emb_rel_k = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.emb_rel_v = nn.Parameter(torch.randn(1, window_size * 2 + 1, self.k_channels) * rel_stddev)
        self.conv_o = ConvP(channels, out_channels, 1)
        nn.init.xavier_uniform_(self.conv_q.weight)
        nn.init.xavier_uniform_(self.conv_k.weight)
        if proximal_init:
            self.conv_k.weight.data.copy_(self.conv_q.weight.data)
            self.conv_k.bias.data.copy_(self.conv_q.bias.data)
        nn.init.xavier_uniform_(self.conv_v.weight)
        # q, k, v projections run as ONE GEMM: [R, C] x [C, 3C] forward, K = 3C for the data gradient
        self._pc_qkv = None
        self.qkv_bias = None
        for i, cv in enumerate((self.conv_q, self.conv_k, self.conv_v)):
            cv.cat_slice = (lambda i=i: PackSliceN(self.pc_qkv, i * self.channels, self.channels, self.channels))

    @property
    def pc_qkv(self):
        dev = self.conv_q.weight.device
        if self._pc_qkv is None or self._pc_qkv.fwd.device != dev:
            self._pc_qkv = PackedConv(3 * self.channels, self.channels, 1, False, device=dev)
        return self._pc_qkv

    def _refresh_padded(self):
        with torch.no_grad():
            self.qkv_bias = torch.cat([self.conv_q.bias, self.conv_k.bias, self.conv_v.bias])

    def forward(self, x, c, attn_mask=None):
        """x is c (self-attention); attn_mask [b,1,t,t] = x_mask ⊗ x_mask as attentions.py:61 builds it."""
        assert x is c, "relative attention is only available for self-attention (attentions.py:250)"
        prepare_all(self)
        x_mask = attn_mask[:, :, :, 0].clone() if attn_mask is not None else torch.ones_like(x[:, :1])
        x_mask = (attn_mask.amax(dim=3) > 0).to(x.dtype) if attn_mask is not None else x_mask
        runner = _MHARunner(self, x_mask, self.training)
        out, p = _RowsFn.apply(runner, 1, x, *runner.params)
        self.attn = p
        return out


class _MHARunner:
    def __init__(self, att, x_mask, train, seed=0):
        self.att, self.x_mask, self.train, self.seed = att, x_mask, train, seed
        self.params = list(att.parameters())

    def forward(self, x, *_):
        B, C, T = x.shape
        rc = RowsCtx(_mask_lengths(self.x_mask), T)
        xb = rc.to_rows(x.detach() * self.x_mask, torch.bfloat16)
        y, saved = encoder_impl.mha_fwd(rc, self.att, xb, self.att.p_dropout if self.train else 0.0, self.seed)
        return (rc.from_rows(y, torch.float32), saved[5]), (rc, saved)

    def backward(self, saved_all, dy, _dp):
        rc, saved = saved_all
        grads = {}
        with wgrad.WgradQueue(dy.device, site=self.att):
            dxb = encoder_impl.mha_bwd(rc, self.att, saved, rc.to_rows(dy, torch.bfloat16), grads)
        return [rc.from_rows(dxb, torch.float32) * self.x_mask] + [grads.get(p) for p in self.params]


class FFN(nn.Module):
    """reference attentions.FFN (attentions.py:350-372), relu activation."""

    def __init__(self, in_channels, out_channels, filter_channels, kernel_size, p_dropout=0., activation=None):
        super().__init__()
        assert activation is None, "the reference configs use relu"
        self.in_channels, self.out_channels, self.filter_channels = in_channels, out_channels, filter_channels
        self.kernel_size, self.p_dropout, self.activation = kernel_size, p_dropout, activation
        self.conv_1 = ConvP(in_channels, filter_channels, kernel_size)
        self.conv_2 = ConvP(filter_channels, out_channels, kernel_size)


class Encoder(nn.Module):
    """reference attentions.Encoder (attentions.py:12-86): post-LN transformer with conv FFN."""

    def __init__(self, hidden_channels, filter_channels, n_heads, n_layers, kernel_size=1, p_dropout=0., window_size=None,
                 block_length=None, gin_channels=0, emoin_channels=0, **kwargs):
        super().__init__()
        self.hidden_channels, self.filter_channels, self.n_heads, self.n_layers = hidden_channels, filter_channels, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.window_size, self.block_length, self.gin_channels = \
            kernel_size, p_dropout, window_size, block_length, gin_channels
        self.attn_layers = nn.ModuleList()
        self.norm_layers_1 = nn.ModuleList()
        self.ffn_layers = nn.ModuleList()
        self.norm_layers_2 = nn.ModuleList()
        for _ in range(n_layers):
            self.attn_layers.append(MultiHeadAttention(hidden_channels, hidden_channels, n_heads, window_size=window_size,
                                                       p_dropout=p_dropout, block_length=block_length))
            self.norm_layers_1.append(LayerNorm(hidden_channels))
            self.ffn_layers.append(FFN(hidden_channels, hidden_channels, filter_channels, kernel_size, p_dropout=p_dropout))
            self.norm_layers_2.append(LayerNorm(hidden_channels))
        if gin_channels != 0:
            self.cond_g = nn.Linear(gin_channels, hidden_channels)

    COND_LAYER = 2          # `if i == 3 - 1 and g is not None` (attentions.py:66)

    def cond_vec(self, g):
        """cond_g(g) (attentions.py:67) for g [b,gin,1]: a [B,gin]x[gin,H] product on B rows — host-side PyTorch
        plumbing, differentiable w.r.t. g and cond_g's parameters; the kernels add it to the rows."""
        if g is None or self.n_layers <= self.COND_LAYER:
            return None
        return torch.nn.functional.linear(g.squeeze(-1), self.cond_g.weight, self.cond_g.bias)

    def forward(self, x, x_mask, g=None, emo=None):
        assert emo is None, "emotion conditioning is commented out in the reference encoder (attentions.py:69-70)"
        prepare_all(self)
        vec = self.cond_vec(g)
        runner = _EncoderRunner(self, x_mask, self.training, has_cond=vec is not None)
        (out,) = _RowsFn.apply(runner, 1, x, *([vec] if vec is not None else []), *runner.params)
        return out


class _EncoderRunner:
    def __init__(self, enc, x_mask, train, seed=0, has_cond=False):
        self.enc, self.x_mask, self.train, self.seed, self.has_cond = enc, x_mask, train, seed, has_cond
        self.params = [p for n, p in enc.named_parameters() if not n.startswith("cond_g.")]

    def forward(self, x, *rest):
        from . import ops
        vec = rest[0] if self.has_cond else None
        B, C, T = x.shape
        rc = RowsCtx(_mask_lengths(self.x_mask), T)
        xm = x.detach().float() * self.x_mask
        xr, xb = rc.to_rows(xm), rc.to_rows(xm, torch.bfloat16)
        saved = []
        for i in range(self.enc.n_layers):
            if i == self.enc.COND_LAYER and vec is not None:
                xr, xb = ops.rows_add_cond(rc, xr, None, vec)
            xr, xb, s = encoder_impl.layer_fwd(rc, self.enc, i, xr, xb, self.train, self.seed + 8 * i)
            saved.append(s)
        return (rc.from_rows(xr),), (rc, saved)

    def backward(self, saved_all, dout):
        from . import ops
        rc, saved = saved_all
        grads = {}
        dvec = None
        dx, dxb = rc.to_rows(dout.float() * self.x_mask), None
        with wgrad.WgradQueue(dout.device, site=self.enc):
            for i in reversed(range(self.enc.n_layers)):
                dx, dxb = encoder_impl.layer_bwd(rc, self.enc, i, saved[i], dx, dxb, grads)
                if i == self.enc.COND_LAYER and self.has_cond:
                    dvec = ops.cond_grad(rc, dx, dxb)
        tot = rc.from_rows(dx) + rc.from_rows(dxb, torch.float32)
        return [tot * self.x_mask] + ([dvec] if self.has_cond else []) + [grads.get(p) for p in self.params]
