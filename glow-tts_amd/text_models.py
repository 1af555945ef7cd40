"""Host-side mirror of the reference's models.py, text side + training glue.

  DurationPredictor   reference models.py:560-612
  TextEncoder         reference models.py:614-716
  FlowGenerator       reference models.py:792-1133 — the upstream-equivalent live sub-graph of
                      forward() for the base configs (the fork's own FlowGenerator
                      only constructs for base_blank_emo_lang_pitch.json, which is outside the
                      round-1 scope).
"""
import math
import contextlib
import os

import torch
from torch import nn

from . import _lib, encoder_impl, ops
from . import monotonic_align
from .attentions import Encoder
from .modules import ConvP, ConvReluNorm, LayerNorm, _RowsFn, prepare_all
from .ops import HALO, PackedConv, RowsCtx


class _PaddedConv:
    """An [1, C, 1] projection run as an 8-channel conv (MFMA tiles want N % 8 == 0).  The padded
    weight/bias live in persistent buffers refreshed in place, so the pack table's pointers stay valid."""
    weight_norm = False

    def __init__(self, conv):
        self.conv, self.pc, self.weight, self.bias = conv, None, None, None

    def refresh(self):
        c = self.conv
        dev = c.weight.device
        if self.pc is None or self.pc.fwd.device != dev:
            self.pc = PackedConv(8, c.in_channels, c.kernel_size, False, device=dev)
            self.weight = torch.zeros(8, c.in_channels, c.kernel_size, device=dev)
            self.bias = torch.zeros(8, device=dev)
        self.weight[:1].copy_(c.weight.detach())
        self.bias[:1].copy_(c.bias.detach())

    def prepare(self):
        self.refresh()
        self.pc.pack(self.weight, None)


class DurationPredictor(nn.Module):
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, gin_channels=0, lin_channels=0, emoin_channels=0):
        super().__init__()
        assert emoin_channels == 0, "emotion conditioned predictor: the reference never builds it (models.py:671)"
        self.in_channels, self.filter_channels, self.kernel_size, self.p_dropout = in_channels, filter_channels, kernel_size, p_dropout
        self.gin_channels, self.lin_channels = gin_channels, lin_channels
        self.conv_1 = ConvP(in_channels, filter_channels, kernel_size)
        self.norm_1 = LayerNorm(filter_channels)
        self.conv_2 = ConvP(filter_channels, filter_channels, kernel_size)
        self.norm_2 = LayerNorm(filter_channels)
        self.proj = ConvP(filter_channels, 1, 1)
        self.proj.no_pack = True                      # runs through proj_pad (8 output channels); parameters only
        self.proj_pad = _PaddedConv(self.proj)
        if gin_channels != 0:
            self.cond = nn.Conv1d(gin_channels, in_channels, 1)          # models.py:579-580; B rows: host-side plumbing
        if lin_channels != 0:
            self.cond_lang = nn.Conv1d(lin_channels, in_channels, 1)     # models.py:582-583

    def cond_vec(self, g, l=None):
        """cond(detach(g)) + cond_lang(detach(l)) (models.py:587-597) for g [b,gin,1], l [b,lin,1] -> [B, in_channels];
        the kernels add it to the rows."""
        F = torch.nn.functional
        v = None
        if g is not None:
            v = F.linear(g.detach().squeeze(-1), self.cond.weight.squeeze(-1), self.cond.bias)
        if l is not None:
            vl = F.linear(l.detach().squeeze(-1), self.cond_lang.weight.squeeze(-1), self.cond_lang.bias)
            v = vl if v is None else v + vl
        return v

    def _refresh_padded(self):
        self.proj_pad.refresh()

    def _pack_entries_extra(self):
        return [(self.proj_pad.weight, None, self.proj_pad.pc)]

    def prepare_extra(self):
        pass


class TextEncoder(nn.Module):
    def __init__(self, n_vocab, out_channels, hidden_channels, filter_channels, filter_channels_dp, n_heads, n_layers,
                 kernel_size, p_dropout, window_size=None, block_length=None, mean_only=False, prenet=False, use_sdp=False,
                 gin_channels=0, lin_channels=0, emoin_channels=0):
        super().__init__()
        self.use_sdp, self.gin_channels, self.lin_channels = use_sdp, gin_channels, lin_channels
        self.n_vocab, self.out_channels, self.hidden_channels = n_vocab, out_channels, hidden_channels
        self.filter_channels, self.filter_channels_dp, self.n_heads, self.n_layers = filter_channels, filter_channels_dp, n_heads, n_layers
        self.kernel_size, self.p_dropout, self.window_size, self.mean_only, self.prenet = kernel_size, p_dropout, window_size, mean_only, prenet
        # multi-language models: the token embedding is lin_channels narrower and the language embedding fills the rest
        # of every position (models.py:654-664, 698-699)
        self.emb = nn.Embedding(n_vocab, hidden_channels - lin_channels)
        nn.init.normal_(self.emb.weight, 0.0, (hidden_channels - lin_channels) ** -0.5)
        if use_sdp:                                                  # models.py:668-670
            from .predictors import StochasticDurationPredictor
            self.proj_w = StochasticDurationPredictor(hidden_channels, 192, 3, 0.5, 4, gin_channels=gin_channels, lin_channels=lin_channels)
        else:
            self.proj_w = DurationPredictor(hidden_channels, filter_channels_dp, kernel_size, p_dropout, gin_channels=gin_channels,
                                            lin_channels=lin_channels)
        if prenet:
            self.pre = ConvReluNorm(hidden_channels, hidden_channels, hidden_channels, kernel_size=5, n_layers=3, p_dropout=0.5)
        self.encoder = Encoder(hidden_channels, filter_channels, n_heads, n_layers, kernel_size, p_dropout,
                               window_size=window_size, block_length=block_length, gin_channels=gin_channels)
        self.proj_m = ConvP(hidden_channels, out_channels, 1)
        if not mean_only:
            self.proj_s = ConvP(hidden_channels, out_channels, 1)
        self._step = 0

    def forward(self, x, x_lengths, l=None, g=None, emo=None, prepared=False):
        """ids [b, t] int64, lengths [b] -> (x [b,H,t], x_m [b,80,t], x_logs [b,80,t], x_mask [b,1,t])
        exactly as reference models.py:692-716."""
        assert emo is None, "emotion conditioning is commented out in the reference encoder (models.py:695-696)"
        assert (l is None) == (self.lin_channels == 0), "l [b, lin_channels, 1] is required exactly when lin_channels != 0"
        if not prepared:
            prepare_all(self)
        self._step += 1
        T = x.shape[1]
        x_mask = ops.length_mask(x_lengths, T)
        vec = self.encoder.cond_vec(g)                # speaker vector, added before encoder layer index 2
        lvec = None if l is None else l.squeeze(-1)     # [b, lin]: a differentiable input of the node (emb_l upstream)
        runner = _TextEncoderRunner(self, x, x_lengths, self.training, seed=(self._step * 104729) & 0x7fffffff,
                                    has_cond=vec is not None, has_lang=lvec is not None)
        outs = _RowsFn.apply(runner, 3, *([vec] if vec is not None else []), *([lvec] if lvec is not None else []), *runner.params)
        xo, x_m, x_logs = outs[0], outs[1], outs[2]
        self._last_rows = runner.last            # (rc, xb_final) for the duration predictor
        return xo, x_m, x_logs, x_mask


class _TextEncoderRunner:
    def __init__(self, te, ids, lengths, train, seed, has_cond=False, has_lang=False):
        self.te, self.ids, self.lengths, self.train, self.seed = te, ids.contiguous(), lengths, train, seed
        self.has_cond, self.has_lang = has_cond, has_lang
        self.params = [p for n, p in te.named_parameters() if not n.startswith("proj_w.") and not n.startswith("encoder.cond_g.")]
        self.last = None

    def forward(self, *rest):
        L = _lib.lib()
        te = self.te
        vec = rest[0] if self.has_cond else None
        lvec = rest[int(self.has_cond)] if self.has_lang else None
        B, T = self.ids.shape
        dev = self.ids.device
        C = te.hidden_channels
        Ce = C - te.lin_channels                         # token-embedding channels; the language vector fills [Ce, C)
        rc = ops.make_ctx(self.lengths.to(torch.int32), T, "x", cfg=getattr(te, "rows_cfg", None))
        x = torch.empty(rc.R, C, dtype=torch.float32, device=dev)
        xb = torch.empty(rc.R, C, dtype=torch.bfloat16, device=dev)
        emb = te.emb.weight.detach()
        _lib.check(L.gt_embedding_fwd(_lib.ptr(self.ids), _lib.ptr(emb), _lib.ptr(rc.lengths), _lib.ptr(x), _lib.ptr(xb),
                                      B, T, rc.Tp, _lib.ptr(rc.row0), rc.R, Ce, C, math.sqrt(C), _lib.current_stream(dev)), "gt_embedding_fwd")
        if lvec is not None:                             # x = cat(emb * sqrt(H), l expanded over time) (models.py:698-699)
            lc = lvec.detach().float().contiguous()
            zero = ops.zeros_small((rc.R, te.lin_channels), torch.float32, dev)
            _lib.check(L.gt_rows_add_cond(_lib.ptr(zero), te.lin_channels, None, 0, _lib.ptr(lc), _lib.ptr(rc.rowmask),
                                          x.data_ptr() + 4 * Ce, C, xb.data_ptr() + 2 * Ce, C, B, rc.R, te.lin_channels, rc.Tp,
                                          _lib.ptr(rc.row0), _lib.current_stream(dev)), "gt_rows_add_cond")
        s_pre = None
        if te.prenet:
            x, xb, s_pre = encoder_impl.crn_fwd(rc, te.pre, x, xb, self.train, self.seed)
        s_layers = []
        for i in range(te.encoder.n_layers):
            if i == te.encoder.COND_LAYER and vec is not None:
                x, xb = ops.rows_add_cond(rc, x, None, vec)
            x, xb, s = encoder_impl.layer_fwd(rc, te.encoder, i, x, xb, self.train, self.seed + 16 + 8 * i)
            s_layers.append(s)
        from .ops import conv_rows
        xm_r = conv_rows(xb, te.proj_m.pc, rc, bias=te.proj_m.bias, mask=True, out_f32=True)
        xs_r = None if te.mean_only else conv_rows(xb, te.proj_s.pc, rc, bias=te.proj_s.bias, mask=True, out_f32=True)
        xo = rc.from_rows(x)
        x_m = rc.from_rows(xm_r)
        x_logs = torch.zeros_like(x_m) if xs_r is None else rc.from_rows(xs_r)
        self.last = (rc, xb)
        return (xo, x_m, x_logs), (rc, s_pre, s_layers, xb)

    def backward(self, saved_all, dxo, dx_m, dx_logs):
        L = _lib.lib()
        from .flow_impl import conv_param_grads
        from .ops import conv_rows
        rc, s_pre, s_layers, xb_final = saved_all
        te = self.te
        dev = rc.device
        C = te.hidden_channels
        grads = {}
        dxb = None
        dvec = None
        if dxo is None and dx_m is None and (te.mean_only or dx_logs is None):
            return [None] * (len(self.params) + int(self.has_cond) + int(self.has_lang))
        from . import wgrad
        ops.mark("enc bwd begin")
        with wgrad.WgradQueue(dev, site=te):
            if dx_m is not None:
                d = rc.to_rows(dx_m.float(), torch.bfloat16)
                grads.update(conv_param_grads(te.proj_m, xb_final, d, rc.R))
                dxb = conv_rows(d, te.proj_m.pc, rc, dgrad=True)
            if not te.mean_only and dx_logs is not None:
                d2 = rc.to_rows(dx_logs.float(), torch.bfloat16)
                grads.update(conv_param_grads(te.proj_s, xb_final, d2, rc.R))
                dxb = conv_rows(d2, te.proj_s.pc, rc, dgrad=True, addend=dxb)
            dx = rc.to_rows(dxo.float()) if dxo is not None else None
            if dx is None:
                dx = torch.zeros(rc.R, C, dtype=torch.float32, device=dev)
            for i in reversed(range(te.encoder.n_layers)):
                dx, dxb = encoder_impl.layer_bwd(rc, te.encoder, i, s_layers[i], dx, dxb, grads)
                if i == te.encoder.COND_LAYER and self.has_cond:
                    dvec = ops.cond_grad(rc, dx, dxb)
            if te.prenet:
                dx, dxb = encoder_impl.crn_bwd(rc, te.pre, s_pre, dx, dxb, grads)
            ops.mark("enc dgrad end")
        ops.mark("enc wgrad end")
        tot, _ = encoder_impl._sum_grads_to_bf16(rc, dx, dxb, C)
        demb = ops.grad_accumulator(te.emb.weight)
        B, T = self.ids.shape
        Ce = C - te.lin_channels
        _lib.check(L.gt_embedding_bwd(_lib.ptr(self.ids), _lib.ptr(tot), _lib.ptr(rc.lengths), _lib.ptr(demb), B, T, rc.Tp, _lib.ptr(rc.row0), rc.R, Ce, C,
                                      math.sqrt(C), _lib.current_stream(dev)), "gt_embedding_bwd")
        grads[te.emb.weight] = demb
        dl = []
        if self.has_lang:                                # the language vector was broadcast over time: its gradient is the row sum
            dl = [rc.utt_sum(tot[:, Ce:], torch.empty(B, te.lin_channels, dtype=torch.float32, device=dev))]
        return ([dvec] if self.has_cond else []) + dl + [grads.get(p) for p in self.params]


class _DurationRunner:
    """logw = DurationPredictor(x.detach(), x_mask) as one autograd node over its own parameters."""

    def __init__(self, dp, rc, xb, train, seed, has_cond=False):
        self.dp, self.rc, self.xb, self.train, self.seed, self.has_cond = dp, rc, xb, train, seed, has_cond
        self.params = [p for n, p in dp.named_parameters() if not n.startswith("cond.")]

    def forward(self, *rest):
        rc = self.rc
        xb = self.xb
        if self.has_cond:                                   # x + cond(g), then * x_mask (models.py:587-589, 598)
            _, xb = ops.rows_add_cond(rc, None, xb, rest[0], want_f32=False)
        out, saved = encoder_impl.dp_fwd(rc, self.dp, xb, self.train, self.seed)
        logw = rc.from_rows(out)[:, :1].contiguous()                                              # [b,1,t]
        return (logw,), saved

    def backward(self, saved, dlogw):
        rc = self.rc
        grads = {}
        d8 = torch.zeros(rc.B, 8, rc.T, dtype=torch.float32, device=rc.device)
        d8[:, 0] = dlogw[:, 0].float()
        dout = rc.to_rows(d8)
        from . import wgrad
        ops.mark("dp bwd begin")
        with wgrad.WgradQueue(rc.device, site=self.dp):
            dxb = encoder_impl.dp_bwd(rc, self.dp, saved, dout, grads, want_dx=self.has_cond)
        ops.mark("dp bwd end")
        return ([ops.cond_grad(rc, dxb)] if self.has_cond else []) + [grads.get(p) for p in self.params]


class _LogpMasFn:
    """no_grad block of models.py:1076-1083: logp lattice + MAS on the device."""

    @staticmethod
    def run(x_m, x_logs, z, x_lengths, y_lengths, mean_only):
        L = _lib.lib()
        B, C, Tx = x_m.shape
        Ty = z.shape[2]
        dev = z.device
        logp = torch.empty(B, Tx, Ty, dtype=torch.float32, device=dev)
        xm = x_m.detach().float().contiguous()
        xs = None if mean_only else x_logs.detach().float().contiguous()
        zz = z.detach().float().contiguous()
        _lib.check(L.gt_logp_f32(_lib.ptr(xm), _lib.ptr(xs), _lib.ptr(zz), _lib.ptr(logp), B, C, Tx, Ty,
                                 _lib.current_stream(dev)), "gt_logp_f32")
        r = monotonic_align.maximum_path_lengths(logp, x_lengths.to(torch.int32), y_lengths.to(torch.int32),
                                                 want_durations=True, want_frame2token=True, keep_workspace=True)
        return logp, r


class _PriorExpandFn(torch.autograd.Function):
    """z_m = attn^T x_m (models.py:1118) as a gather; backward = segment sums over the MAS intervals."""

    @staticmethod
    def forward(ctx, x_m, f2t, starts):
        L = _lib.lib()
        B, C, Tx = x_m.shape
        Ty = f2t.shape[1]
        xm = x_m.detach().float().contiguous()
        z_m = torch.empty(B, C, Ty, dtype=torch.float32, device=x_m.device)
        _lib.check(L.gt_prior_expand(_lib.ptr(xm), _lib.ptr(f2t), _lib.ptr(z_m), B, C, Tx, Ty, _lib.current_stream(x_m.device)), "gt_prior_expand")
        ctx.f2t, ctx.shape = f2t, (B, C, Tx, Ty)
        return z_m

    @staticmethod
    def backward(ctx, dz_m):
        L = _lib.lib()
        B, C, Tx, Ty = ctx.shape
        d = dz_m.float().contiguous()
        dx_m = torch.empty(B, C, Tx, dtype=torch.float32, device=d.device)
        _lib.check(L.gt_prior_expand_bwd(_lib.ptr(d), _lib.ptr(ctx.f2t), _lib.ptr(dx_m), B, C, Tx, Ty, _lib.current_stream(d.device)),
                   "gt_prior_expand_bwd")
        return dx_m, None, None


class _MleLossFn(torch.autograd.Function):
    """commons.mle_loss (commons.py:28-33) with z_logs == None meaning zeros (mean_only)."""

    @staticmethod
    def forward(ctx, z, m, logs, logdet, mask):
        L = _lib.lib()
        dev = z.device
        zc, mc = z.detach().float().contiguous(), m.detach().float().contiguous()
        lc = None if logs is None else logs.detach().float().contiguous()
        acc = torch.empty(2 * 2048, dtype=torch.float32, device=dev)       # GT_MLE_PARTS partial pairs, all written by the kernel
        _lib.check(L.gt_mle_sums(_lib.ptr(zc), _lib.ptr(mc), _lib.ptr(lc), _lib.ptr(acc), zc.numel(), _lib.current_stream(dev)), "gt_mle_sums")
        # the scalar tail in one launch: loss = (acc[0] + 0.5 acc[1] - sum logdet) / denom + 0.5 log 2pi, denom = C * sum(mask)
        ld = logdet.detach().float().contiguous()
        mk = mask.detach().float().contiguous()
        out = torch.empty(2, dtype=torch.float32, device=dev)
        _lib.check(L.gt_mle_finish(_lib.ptr(acc), _lib.ptr(ld), _lib.ptr(mk), mk.numel(), ld.numel(), z.shape[1], _lib.ptr(out),
                                   _lib.current_stream(dev)), "gt_mle_finish")
        ctx.saved = (zc, mc, lc, out, logdet.shape)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        zc, mc, lc, out, ld_shape = ctx.saved
        dev = zc.device
        gs = g.float().reshape(1).contiguous()                   # divided by the denominator (out[1]) inside the kernel
        dz = torch.empty_like(zc)
        dm = torch.empty_like(mc)
        dl = None if lc is None else torch.empty_like(lc)
        nb = 1
        for d in ld_shape:
            nb *= int(d)
        dlogdet = torch.empty(ld_shape, dtype=torch.float32, device=dev)
        _lib.check(L.gt_mle_bwd(_lib.ptr(zc), _lib.ptr(mc), _lib.ptr(lc), _lib.ptr(gs), _lib.ptr(dz), _lib.ptr(dm), _lib.ptr(dl),
                                zc.numel(), out[1:].data_ptr(), _lib.ptr(dlogdet), nb, _lib.current_stream(dev)), "gt_mle_bwd")
        return dz, dm, dl, dlogdet, None


class _DurationLossFn(torch.autograd.Function):
    """models.py:1089-1092: l_length[b] = sum_t (logw - log(w + 1e-8) * x_mask)^2 / sum(x_mask) in one launch (w: MAS durations,
    no gradient); backward one more."""

    @staticmethod
    def forward(ctx, logw, w, x_lengths):
        L = _lib.lib()
        B, _, Tx = logw.shape
        lw = logw.detach().float().contiguous()
        wc = w.detach().float().contiguous()
        xl = x_lengths.to(torch.int32).contiguous()
        out = torch.empty(B, dtype=torch.float32, device=lw.device)
        _lib.check(L.gt_duration_loss_fwd(_lib.ptr(lw), _lib.ptr(wc), _lib.ptr(xl), B, Tx, _lib.ptr(out), _lib.current_stream(lw.device)),
                   "gt_duration_loss_fwd")
        ctx.saved = (lw, wc, xl, logw.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        lw, wc, xl, shape = ctx.saved
        B, _, Tx = shape
        gc = g.float().contiguous()
        d = torch.empty(shape, dtype=torch.float32, device=lw.device)
        _lib.check(L.gt_duration_loss_bwd(_lib.ptr(lw), _lib.ptr(wc), _lib.ptr(xl), _lib.ptr(gc), B, Tx, _lib.ptr(d), _lib.current_stream(lw.device)),
                   "gt_duration_loss_bwd")
        return d, None, None


def mle_loss(z, m, logs, logdet, mask):
    """Drop-in for reference commons.mle_loss (commons.py:28-33); logs may be None (== zeros)."""
    return _MleLossFn.apply(z, m, logs, logdet, mask)


# (whether the encoder / the predictors run as parallel branches is per-model state: ops.RowsConfig.encoder_stream / .predictor_branch /
# .energy_on_main of model.rows_cfg)
_ENC_STREAMS = {}            # device -> the branch's stream (one per device and process: streams are a device resource, not model state)


_FRONT_STREAMS = {}


def _front_stream(dev):
    key = str(dev)
    if key not in _FRONT_STREAMS:
        _FRONT_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _FRONT_STREAMS[key]


def _encoder_stream(dev):
    key = str(dev)
    if key not in _ENC_STREAMS:
        _ENC_STREAMS[key] = torch.cuda.Stream(device=dev)
    return _ENC_STREAMS[key]


class FlowGenerator(nn.Module):
    """reference models.FlowGenerator (models.py:792-1256): constructor arguments, forward / infer signatures, return
    structure and state_dict keys of the fork.

    * configs/base_blank_emo_lang_pitch.json (cfg 5, the one config the reference class constructs for — SURVEY F1):
      `FlowGenerator(n_vocab, out_channels=80, n_lang=10, **hps.model)` builds the speaker / emotion front end
      (use_spk_embeds, use_emo_embeds: emb_g, emo_*, elevation / azimuth tables, models.py:904-937), the language
      embedding, the StochasticDurationPredictor inside the encoder (use_sdp) and the stochastic pitch / energy predictors
      (use_spp / use_sep); forward(x, x_lengths, y, y_lengths, g, emo, emo_cartesian, pitch, energy, l) is called exactly as
      train_ms_emo_lang_pitch.py:284-289 calls it and returns the 5-tuple of models.py:1133 with l_pitch / l_energy filled.
    * the base configs (cfg 1-4), for which the fork's class raises NameError (F1): the upstream-equivalent live
      sub-graph — use_sdp=False gives the deterministic DurationPredictor, g (if gin_channels) enters as [b, gin, 1] at
      the encoder / duration-predictor / decoder boundary, entries this sub-graph does not produce are None.

    with_prosody_wn (ours; implied by use_spp / use_sep): create the fork's wn_pitch / wn_energy in every coupling block.
    The reference always creates them (two thirds of the decoder's parameters are dead weight in the base
    configs); here they exist only when something can feed them."""

    def __init__(self, n_vocab, hidden_channels, filter_channels, filter_channels_dp, out_channels, kernel_size=3, n_heads=2,
                 n_layers_enc=10, p_dropout=0., n_blocks_dec=12, kernel_size_dec=5, dilation_rate=5, n_block_layers=4,
                 p_dropout_dec=0., n_speakers=0, n_lang=0, gin_channels=0, lin_channels=0, emoin_channels=0, n_split=4, n_sqz=1,
                 sigmoid_scale=False, window_size=None, block_length=None, mean_only=False, hidden_channels_enc=None,
                 hidden_channels_dec=None, prenet=False, use_spk_embeds=False, use_lang_embeds=False, use_emo_embeds=False,
                 use_sdp=True, use_spp=False, use_sep=False, with_prosody_wn=None, **kwargs):
        super().__init__()
        from .models import FlowSpecDecoder
        self.n_vocab, self.hidden_channels, self.out_channels = n_vocab, hidden_channels, out_channels
        self.n_sqz, self.mean_only, self.gin_channels, self.lin_channels, self.n_lang = n_sqz, mean_only, gin_channels, lin_channels, n_lang
        self.use_spk_embeds, self.use_emo_embeds = use_spk_embeds, use_emo_embeds
        self.use_lang_embeds = use_lang_embeds or (n_lang > 1 and lin_channels > 0)
        self.use_sdp, self.use_spp, self.use_sep = use_sdp, use_spp, use_sep
        henc = hidden_channels_enc or hidden_channels
        self.encoder = TextEncoder(n_vocab, out_channels, henc, filter_channels, filter_channels_dp, n_heads, n_layers_enc,
                                   kernel_size, p_dropout, window_size=window_size, block_length=block_length, mean_only=mean_only,
                                   prenet=prenet, use_sdp=use_sdp, gin_channels=gin_channels, lin_channels=lin_channels)
        if with_prosody_wn is None:
            with_prosody_wn = bool(use_spp or use_sep)
        self.decoder = FlowSpecDecoder(out_channels, hidden_channels_dec or hidden_channels, kernel_size_dec, dilation_rate,
                                       n_blocks_dec, n_block_layers, p_dropout=p_dropout_dec, n_split=n_split, n_sqz=n_sqz,
                                       sigmoid_scale=sigmoid_scale, gin_channels=gin_channels, with_prosody_wn=with_prosody_wn)
        if use_spk_embeds:                                           # models.py:904-906
            self.emb_g = nn.Linear(512, gin_channels // 2)
        if self.use_lang_embeds:                                     # models.py:913-916
            self.emb_l = nn.Embedding(n_lang, lin_channels)
            nn.init.xavier_uniform_(self.emb_l.weight)
        if use_emo_embeds:                                           # models.py:918-937 ("Cartesian Emo")
            q, e = gin_channels // 4, gin_channels // 8
            self.emo_id_proj = nn.Embedding(5, q)
            nn.init.normal_(self.emo_id_proj.weight, mean=0, std=gin_channels // 4 ** -0.5)     # (sic: // binds first)
            self.emo_proj = nn.Linear(q, q, bias=True)
            self.emo_VAD_inten_proj = nn.Linear(1, gin_channels // 2)
            self.elevation_bins = nn.Parameter(torch.linspace(math.pi / 2, math.pi, 2), requires_grad=False)
            self.elevation_emb = nn.Embedding(2, e)
            nn.init.normal_(self.elevation_emb.weight, mean=0, std=gin_channels // 8 ** -0.5)
            self.azimuth_bins = nn.Parameter(torch.linspace(-math.pi / 2, math.pi, 4), requires_grad=False)
            self.azimuth_emb = nn.Embedding(4, e)
            nn.init.normal_(self.azimuth_emb.weight, mean=0, std=gin_channels // 8 ** -0.5)
            self.sty_proj = nn.Linear(q, q, bias=True)
            self.emosty_layer_norm = nn.LayerNorm(gin_channels // 2)
        if use_spp:                                                  # models.py:954-965
            from .predictors import StochasticPitchPredictor
            self.proj_pitch = StochasticPitchPredictor(henc, 256, 3, 0.1, 4, gin_channels=gin_channels)
        if use_sep:                                                  # models.py:970-981
            from .predictors import StochasticEnergyPredictor
            self.proj_energy = StochasticEnergyPredictor(henc, 256, 3, 0.1, 4, gin_channels=gin_channels)
        self._step = 0
        # rows-layout state of THIS model (ragged packing, row rounding, the batch's host-side lengths): shared with the
        # encoder / decoder runners; train.Trainer configures it — nothing process-global
        self.rows_cfg = self.encoder.rows_cfg = self.decoder.rows_cfg = ops.RowsConfig()

    # ---- conditioning front end (models.py:1008-1042) -----------------------------------------------------------------
    def condition(self, g, emo, emo_cartesian):
        """-> the conditioning vector [b, gin, 1] that encoder / predictors / decoder see.  These are ~15 small ops on
        [B, <= 512] tensors per step (B rows): host-side PyTorch plumbing, differentiable w.r.t. all their parameters."""
        F = torch.nn.functional
        if self.use_spk_embeds:
            if g is None:
                raise ValueError("this model owns emb_g: forward needs the raw speaker embedding g [b, 512]")
            g = self.emb_g(F.normalize(g.squeeze(-1) if g.dim() == 3 else g))
        if not self.use_emo_embeds:
            if emo is not None or emo_cartesian is not None:
                raise ValueError("emo / emo_cartesian given to a model built without use_emo_embeds")
            if g is None:
                return None
            return g.unsqueeze(-1) if g.dim() == 2 else g
        if emo is None or emo_cartesian is None or g is None:
            raise ValueError("use_emo_embeds: forward needs g, emo [b] and emo_cartesian [b, 3] (models.py:1018-1042)")
        emos_proj = self.emo_proj(self.emo_id_proj(emo))
        intens = self.emo_VAD_inten_proj(emo_cartesian[:, :1])
        # bucketize can return len(bins) for a coordinate above the last edge — one past the embedding table (the reference
        # then fails with an index error / a device-side fault); such out-of-range coordinates share the last bucket here
        ele = self.elevation_emb(torch.bucketize(emo_cartesian[:, 1].contiguous(), self.elevation_bins).clamp_(max=self.elevation_emb.num_embeddings - 1))
        azi = self.azimuth_emb(torch.bucketize(emo_cartesian[:, 2].contiguous(), self.azimuth_bins).clamp_(max=self.azimuth_emb.num_embeddings - 1))
        style = self.sty_proj(torch.cat((ele, azi), dim=-1))
        emosty = self.emosty_layer_norm(F.softplus(torch.cat((emos_proj, style), dim=-1)))
        return torch.cat((g, intens + emosty), dim=-1).unsqueeze(-1)

    def store_inverse(self):
        """models.py:1255-1256: freeze the model for synthesis (weights packed once, flow scalars cached)."""
        self.prepare()
        self.decoder.store_inverse()

    @torch.no_grad()
    def infer(self, x, x_lengths, y=None, y_lengths=None, g=None, emo=None, emo_cartesian=None, l=None, gst_token=None,
              noise_scale=1., noise_scale_w=1., f0_noise_scale=1., energy_noise_scale=1., length_scale=1., pitch_scale=1.0,
              energy_scale=1.0):
        """Synthesis (reference FlowGenerator.infer, models.py:1135-1231): text -> durations (deterministic predictor, or the
        stochastic one run in reverse) -> expanded prior -> [predicted pitch / energy] -> z = z_m + noise ->
        decoder(reverse=True) -> mel.  Returns ((y, z_m, z_logs, None, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_),
        (pitch, energy)).  The output length is data dependent, so this reads the predicted lengths back from the device once."""
        if self.decoder._inv_cache is None:
            self.prepare()
        self.rows_cfg.host_lengths.clear()
        g = self.condition(g, emo, emo_cartesian)
        if l is not None:
            l = self.emb_l(l).unsqueeze(-1)
        xo, x_m, x_logs, x_mask = self.encoder(x, x_lengths, l=l, g=g, prepared=True)
        rc, xb = self.encoder._last_rows
        pw = self.encoder.proj_w
        dvec = pw.cond_vec(g, l)
        if self.use_sdp:
            nz = torch.randn(rc.R, 2, dtype=torch.float32, device=x.device) * noise_scale_w
            logw = rc.from_rows(pw._reverse_rows(rc, xb, dvec, nz)[:, None].contiguous())
        else:
            runner = _DurationRunner(pw, rc, xb, False, 0, has_cond=dvec is not None)
            (logw,), _ = runner.forward(*([dvec] if dvec is not None else []))
        w = torch.exp(logw) * x_mask * length_scale
        w_ceil = torch.ceil(w)
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()
        Ty = int(y_lengths.max().item())
        z_mask = (torch.arange(Ty, device=x.device)[None, :] < y_lengths[:, None]).unsqueeze(1).to(x_mask.dtype)
        # commons.generate_path (commons.py:127-143): token i owns frames [cum_i - d_i, cum_i)
        dur = w_ceil.squeeze(1)
        cum = torch.cumsum(dur, 1)
        j = torch.arange(Ty, device=x.device, dtype=dur.dtype)
        attn = ((j[None, None, :] < cum[:, :, None]) & (j[None, None, :] >= (cum - dur)[:, :, None])).to(x_mask.dtype)
        attn = (attn * x_mask.transpose(1, 2) * z_mask).unsqueeze(1)
        frame2token = torch.searchsorted(cum.contiguous(), j[None, :].expand(cum.shape[0], Ty).contiguous(), right=True)
        frame2token = frame2token.clamp_(max=dur.shape[1] - 1).to(torch.int32).contiguous()
        L = _lib.lib()
        B, C, Tx = x_m.shape
        xm = x_m.float().contiguous()
        z_m = torch.empty(B, C, Ty, dtype=torch.float32, device=x.device)
        _lib.check(L.gt_prior_expand(_lib.ptr(xm), _lib.ptr(frame2token), _lib.ptr(z_m), B, C, Tx, Ty, _lib.current_stream(x.device)),
                   "gt_prior_expand")
        z_m = z_m * z_mask
        if self.mean_only:
            z_logs = torch.zeros_like(z_m)
        else:
            xs = x_logs.float().contiguous()
            z_logs = torch.empty_like(z_m)
            _lib.check(L.gt_prior_expand(_lib.ptr(xs), _lib.ptr(frame2token), _lib.ptr(z_logs), B, C, Tx, Ty, _lib.current_stream(x.device)),
                       "gt_prior_expand")
            z_logs = z_logs * z_mask
        logw_ = torch.log(1e-8 + torch.sum(attn.squeeze(1), -1)).unsqueeze(1) * x_mask
        z = (z_m + torch.exp(z_logs) * torch.randn_like(z_m) * noise_scale) * z_mask
        pitch = energy = None
        if self.use_spp or self.use_sep:                              # models.py:1203-1228
            rcf = ops.RowsCtx(y_lengths.to(torch.int32), Ty)
            xf = self._gather_features(rc, xb, rcf, frame2token)
            if self.use_spp:
                nz = torch.randn(rcf.R, 2, dtype=torch.float32, device=x.device) * f0_noise_scale
                pitch = rcf.from_rows(self.proj_pitch._reverse_rows(rcf, xf, self.proj_pitch.cond_vec(g), nz)[:, None].contiguous()).squeeze(1) * pitch_scale
            if self.use_sep:
                nz = torch.randn(rcf.R, 2, dtype=torch.float32, device=x.device) * energy_noise_scale
                energy = rcf.from_rows(self.proj_energy._reverse_rows(rcf, xf, self.proj_energy.cond_vec(g), nz)[:, None].contiguous()).squeeze(1) * energy_scale
        yo, logdet = self.decoder(z, z_mask, g=g, pitch=pitch, energy=energy, reverse=True, prepared=True)
        return (yo, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, logw, logw_), (pitch, energy)

    @staticmethod
    def _gather_features(rcx, xb, rcf, frame2token):
        """x_feature = x @ attn (models.py:1094) for a hard path: frame rows <- token rows (gt_rows_gather_tokens), bf16"""
        C = xb.shape[1]
        out = torch.empty(rcf.R, C, dtype=torch.bfloat16, device=xb.device)
        _lib.check(_lib.lib().gt_rows_gather_tokens(_lib.ptr(xb), xb.stride(0), _lib.ptr(frame2token), frame2token.shape[1],
                                                    _lib.ptr(rcx.row0), rcx.Tp, _lib.ptr(rcf.row_utt()), _lib.ptr(rcf.row0), rcf.Tp,
                                                    _lib.ptr(rcf.rowmask), _lib.ptr(out), rcf.R, C, _lib.current_stream(xb.device)),
                   "gt_rows_gather_tokens")
        return out

    def _predict_logw(self, g, l=None):
        """logw = proj_w(x.detach(), x_mask, g=g, l=l) (models.py:1090) on the rows the text encoder just produced."""
        rc, xb = self.encoder._last_rows
        dvec = self.encoder.proj_w.cond_vec(g, l)
        runner = _DurationRunner(self.encoder.proj_w, rc, xb, self.training, seed=(self._step * 31337) & 0x7fffffff,
                                 has_cond=dvec is not None)
        (logw,) = _RowsFn.apply(runner, 1, *([dvec] if dvec is not None else []), *runner.params)
        return logw

    def backward_encoder(self):
        """Second half of a backward started with defer_encoder_backward=True."""
        pend = [(src, leaf.grad) for src, leaf in self._deferred if leaf.grad is not None]
        self._deferred = []
        if pend:
            torch.autograd.backward([s for s, _ in pend], [g for _, g in pend])

    def prepare(self, side=None, part="all", join=True):
        """Re-pack every conv weight for the MFMA kernels (once per optimizer step).  side: see modules.prepare_all.
        part: "all", or one of the two halves it consists of — "decoder" (the flow decoder's convs: 90 % of the weights; train.Trainer
        packs them at the END of a step, right behind the optimizer's pass over them, beside the text encoder's last backward launches)
        and "rest" (every other child; the head of the next step).  The halves keep their own descriptor tables and derived-bias
        buffers, so a captured step that packs by halves and an eager call that packs everything write the same memory."""
        if part in ("all", "rest"):
            for child in self.children():
                if child is not self.decoder:
                    prepare_all(child, side, join)
            if not self.use_sdp:
                self.encoder.proj_w.prepare_extra()
        if part in ("all", "decoder"):
            prepare_all(self.decoder, side if part == "all" else None, join)

    @staticmethod
    def _contour(c, y_max_length):
        """models.py:1054-1071: raw pitch / energy [b,1,t] (or [b,t]) -> log, with zeros (unvoiced / silent frames) kept 0."""
        if c is None:
            return None
        c = (c.squeeze(1) if c.dim() == 3 else c)[:, :y_max_length]
        zero = c == 0.0
        n = torch.log(torch.clamp(c, min=torch.finfo(c.dtype).tiny))
        return n.masked_fill(zero, 0.0).unsqueeze(1)

    @torch.no_grad()
    def voice_conversion(self, y, y_lengths, spk_embed_src, spk_embed_tgt, l=None):
        """models.py:1233-1247: mel of the source speaker -> latent z through the decoder conditioned on the source speaker's vector
        -> mel through the REVERSE decoder conditioned on the target's.  As in the reference, emb_g is applied to the raw embeddings
        (no F.normalize here, unlike forward) and an optional language id is embedded, normalised and concatenated to both vectors
        (which only a decoder built with gin_channels + lin_channels conditioning inputs accepts)."""
        if not hasattr(self, "emb_g"):
            raise ValueError("voice_conversion needs a model built with use_spk_embeds (emb_g, models.py:1234-1235)")
        if self.decoder._inv_cache is None:
            self.prepare()
        g_src = self.emb_g(spk_embed_src).unsqueeze(-1)
        g_tgt = self.emb_g(spk_embed_tgt).unsqueeze(-1)
        if l is not None:
            lv = torch.nn.functional.normalize(self.emb_l(l)).unsqueeze(-1)
            g_src, g_tgt = torch.cat([g_src, lv], 1), torch.cat([g_tgt, lv], 1)
        z_mask = ops.length_mask(y_lengths, y.shape[2]).to(y.dtype)
        z, _ = self.decoder(y, z_mask, g=g_src, reverse=False, prepared=True)
        y_conv, _ = self.decoder(z, z_mask, g=g_tgt, reverse=True, prepared=True)
        return y_conv

    def preprocess(self, y, y_lengths, y_max_length):
        """reference models.py:1248-1253"""
        if y_max_length is not None:
            y_max_length = (y_max_length // self.n_sqz) * self.n_sqz
            y = y[:, :, :y_max_length]
        y_lengths = torch.div(y_lengths, self.n_sqz, rounding_mode="floor") * self.n_sqz
        return y, y_lengths, y_max_length

    def forward(self, x, x_lengths, y=None, y_lengths=None, g=None, emo=None, emo_cartesian=None, pitch=None, energy=None, l=None,
                lengths_host=None, defer_encoder_backward=False, noise=None, path=None):
        """models.py:1007-1133.  Beyond the reference's arguments:
        lengths_host = (x_lengths, y_lengths) as Python ints: lets the ragged rows layout (self.rows_cfg.ragged) size its
        buffers without a device sync (the data loader has them); without it they are read back from the device.
        defer_encoder_backward: cut the autograd graph at the text encoder's outputs, so that `loss.backward()` yields the
        decoder's (and the predictors') gradients only and `backward_encoder()` runs the rest later — the data-parallel
        trainer all-reduces the decoder's 90 % of the gradient bytes while the encoder's backward runs.
        noise = (e_w [b,2,t_x], e_p [b,1,t_y], e_e [b,1,t_y]): the predictors' torch.randn draws (models.py:288,383,457),
        injected by the parity tests."""
        # The conditioning front end (speaker / emotion / language embeddings: ~40 small launches on [B, <= 512] tensors) runs on a
        # stream of its own when the step forks anyway: autograd then replays its BACKWARD there too — it needs the conditioning
        # vector's gradient from every consumer, so on the caller's stream it queued behind the decoder's weight gradients, optimizer
        # pass and packing and ended the step with 0.65 ms of small launches on an idle machine (cfg 5).
        front = self.rows_cfg.front_stream and self.rows_cfg.encoder_stream and x.is_cuda and torch.is_grad_enabled() and \
            (self.use_spk_embeds or self.use_emo_embeds or l is not None)
        if front:
            caller = torch.cuda.current_stream(x.device)
            fs = _front_stream(x.device)
            fs.wait_stream(caller)
            with torch.cuda.stream(fs):
                g = self.condition(g, emo, emo_cartesian)
                if l is not None:
                    l = self.emb_l(l).unsqueeze(-1)
            caller.wait_stream(fs)
            for t_ in (g, l):
                if t_ is not None:
                    t_.record_stream(caller)
        else:
            g = self.condition(g, emo, emo_cartesian)
            if l is not None:
                l = self.emb_l(l).unsqueeze(-1)                      # language ids [b] -> [b, lin_channels, 1] (models.py:1012-1013)
        pending = self.__dict__.pop("_prepare_side", None)           # train.Trainer packed on this stream and left the join to us:
        if pending is not None:                                      # the front end above ran beside the packing launches
            torch.cuda.current_stream(x.device).wait_stream(pending)
        assert (g is None) == (self.gin_channels == 0), "a speaker / conditioning vector is required exactly when gin_channels != 0"
        if (self.use_spp and pitch is None) or (self.use_sep and energy is None):
            raise ValueError("use_spp / use_sep: forward needs the pitch / energy contours (models.py:1057-1115)")
        if self.__dict__.pop("_prepared_by_trainer", False):
            pass                                                   # train.Trainer._begin packed beside its accumulator fills
        else:
            self.prepare()
        ops.mark("weights packed")
        self._step += 1
        if self.rows_cfg.ragged:
            lh = lengths_host if lengths_host is not None else (x_lengths.tolist(), y_lengths.tolist())
            hl = self.rows_cfg.host_lengths
            hl["x"], hl["y"] = list(lh[0]), list(lh[1])
            hl["f"] = [int(v) // self.n_sqz * self.n_sqz for v in lh[1]]
        else:
            self.rows_cfg.host_lengths.clear()
        # The text encoder (needs the text only) and the decoder (needs the mel only) are independent until the likelihood
        # lattice, and so are their backward passes: the encoder runs on its own stream (a parallel branch of the step's HIP
        # graph) — autograd replays each node's backward on the stream of its forward, so the encoder's backward overlaps
        # the decoder's too.  Both are chains of latency-bound kernels on a fraction of the CUs.
        fork = self.rows_cfg.encoder_stream and x.is_cuda
        logw = None
        # Long texts (cfg 3: T_x = 375 against 436 squeezed mel frames) make the encoder's branch the longer one — its backward ends
        # 0.9 ms after the decoder's —, so the deterministic duration predictor (0.08 ms forward, 0.33 ms backward, needs the encoder's
        # output only and hands nothing back to it: models.py:586 detaches) then runs on the caller's stream instead of the encoder's.
        # Decided on the PADDED shapes, so that it is the same for every batch of a captured graph's key.
        dp_here = fork and not self.use_sdp and self.rows_cfg.dp_balance and x.shape[1] * self.n_sqz >= 0.6 * y.size(2)
        if fork:
            main = torch.cuda.current_stream(x.device)
            enc_stream = _encoder_stream(x.device)
            enc_stream.wait_stream(main)
            with torch.cuda.stream(enc_stream):
                ops.mark("enc fwd begin")
                xo, x_m, x_logs, x_mask = self.encoder(x, x_lengths, l=l, g=g, prepared=True)
                ops.mark("enc fwd end")
                if not self.use_sdp and not dp_here:
                    logw = self._predict_logw(g, l)   # needs the encoder's output only (x is detached, models.py:586): same branch
                    ops.mark("dp fwd end")
        else:
            xo, x_m, x_logs, x_mask = self.encoder(x, x_lengths, l=l, g=g, prepared=True)
        self._deferred = []
        if defer_encoder_backward:
            leaf = x_m.detach().requires_grad_(True)
            self._deferred.append((x_m, leaf)); x_m = leaf
            if not self.mean_only:
                leaf = x_logs.detach().requires_grad_(True)
                self._deferred.append((x_logs, leaf)); x_logs = leaf
        y, y_lengths, y_max_length = self.preprocess(y, y_lengths, y.size(2))
        z_mask = ops.length_mask(y_lengths, y_max_length, x_mask.dtype)
        pitch_norm, energy_norm = self._contour(pitch, y_max_length), self._contour(energy, y_max_length)
        ops.mark("dec fwd begin")
        z, logdet = self.decoder(y, z_mask, g=g, pitch=pitch_norm, energy=energy_norm, prepared=True)
        ops.mark("dec fwd end")
        if fork:
            main.wait_stream(enc_stream)
            ops.mark("fwd joined")
            for t_ in (xo, x_m, x_logs, x_mask, self.encoder._last_rows[1]) + ((logw,) if logw is not None else ()):
                t_.record_stream(main)
            if dp_here:
                logw = self._predict_logw(g, l)
                ops.mark("dp fwd end")
        with torch.no_grad():
            logp, mas = _LogpMasFn.run(x_m, x_logs, z, x_lengths, y_lengths, self.mean_only)
            self._last_mas_path = mas.path                                    # (the searched alignment, whatever `path` says)
            if path is not None:                                              # test hook: a given alignment [b, t_x, t_y] instead
                mas = monotonic_align.result_from_path(path.reshape(mas.path.shape).to(mas.path.dtype), x_lengths.to(torch.int32),
                                                       y_lengths.to(torch.int32))
            attn = mas.path.unsqueeze(1)
        ops.mark("mas done")
        w = mas.durations.unsqueeze(1)                                        # attn.sum(3): models.py:1085
        rcx, xb = self.encoder._last_rows
        # The stochastic predictors need the alignment, but nothing after them does except the loss: they go onto the
        # encoder's stream, so that autograd replays their (long, row-wise) backward there too, beside the decoder's.
        pfork = fork and self.rows_cfg.predictor_branch and (self.use_sdp or self.use_spp or self.use_sep)
        if pfork:
            enc_stream.wait_stream(main)
            for t_ in (w, mas.frame2token, mas.durations, pitch_norm, energy_norm, z_mask, x_mask, y_lengths):
                if t_ is not None:
                    t_.record_stream(enc_stream)
        with (torch.cuda.stream(enc_stream) if pfork else contextlib.nullcontext()):
            l_length, l_pitch, l_energy, logw = self._predictor_losses(rcx, xb, w, x_mask, x_lengths, z_mask, g, l, logw, noise, mas, y_lengths,
                                                                       y_max_length, pitch_norm, energy_norm,
                                                                       energy_stream=main if (pfork and self.rows_cfg.energy_on_main) else None)
        # which stream each loss term was produced on; None = the caller's (joined below, or never forked)
        self._loss_streams = None
        if pfork and self.rows_cfg.join_predictors:
            main.wait_stream(enc_stream)
            for t_ in (l_length, l_pitch, l_energy):
                if t_ is not None:
                    t_.record_stream(main)
        elif pfork:
            # no join: l_length / l_pitch (and l_energy unless its chain ran on the caller's stream) stay on the encoder's stream;
            # the caller seeds the backward per stream (train.Trainer._loss_roots) and joins the streams after it
            self._loss_streams = {"stream": enc_stream, "energy_on_caller": bool(self.rows_cfg.energy_on_main and self.use_spp and self.use_sep)}
        z_m = _PriorExpandFn.apply(x_m, mas.frame2token, mas.workspace)
        z_logs = torch.zeros_like(z_m) if self.mean_only else _PriorExpandFn.apply(x_logs, mas.frame2token, mas.workspace)
        self.last_logp = logp
        return (z, z_m, z_logs, logdet, z_mask), (x_m, x_logs, x_mask), (attn, l_length, l_pitch, l_energy), (None, None, None, None), None

    def _predictor_losses(self, rcx, xb, w, x_mask, x_lengths, z_mask, g, l, logw, noise, mas, y_lengths, y_max_length, pitch_norm, energy_norm,
                          energy_stream=None):
        """l_length, l_pitch, l_energy of models.py:1086-1115 (+ logw of the deterministic duration predictor).
        energy_stream: the energy predictor's chain goes there (the caller's main stream, idle between MAS and the loss) instead of
        queueing behind the duration and pitch predictors on the current (encoder) stream."""
        # The frame-rate features both prosody predictors read come FIRST and the event behind them is what the energy chain on the
        # caller's stream waits for: recorded behind the duration predictor (as until round 3) that chain — and with it the caller's
        # likelihood terms and the decoder's backward — queued behind a predictor it does not depend on.
        l_pitch = l_energy = None
        rcf = xf = zsum = ready = None
        if self.use_spp or self.use_sep:                                       # models.py:1094-1115
            rcf = ops.make_ctx(y_lengths.to(torch.int32), y_max_length, "f", cfg=self.rows_cfg)
            xf = self._gather_features(rcx, xb, rcf, mas.frame2token)
            zsum = torch.sum(z_mask)
            if energy_stream is not None and xf.is_cuda and self.use_spp and self.use_sep:
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(xf.device))           # xf, zsum, the frame-row context: what both chains read
        if self.use_sdp:                                                       # models.py:1086-1088
            pw = self.encoder.proj_w
            w_rows = rcx.to_rows(w.float())[:, 0].contiguous()
            nw = None if noise is None else rcx.to_rows(noise[0].float())
            l_length = pw.nll_rows(rcx, xb, w_rows, pw.cond_vec(g, l), nw) / torch.sum(x_mask)
        else:                                                                  # models.py:1089-1092
            if logw is None:
                logw = self._predict_logw(g, l)
            l_length = _DurationLossFn.apply(logw, w, x_lengths)
        if self.use_spp or self.use_sep:
            if self.use_spp:
                pp = self.proj_pitch
                npz = None if noise is None else rcf.to_rows(noise[1].float())[:, 0].contiguous()
                l_pitch = torch.sum(pp.nll_rows(rcf, xf, rcf.to_rows(pitch_norm.float())[:, 0].contiguous(), pp.cond_vec(g), npz) / zsum)
            if self.use_sep:
                es = energy_stream if ready is not None else None
                if es is not None:
                    es.wait_event(ready)                                    # not the pitch chain queued behind it
                    xf.record_stream(es); zsum.record_stream(es)            # produced on this stream, read on that one
                with (torch.cuda.stream(es) if es is not None else contextlib.nullcontext()):
                    pe = self.proj_energy
                    nez = None if noise is None else rcf.to_rows(noise[2].float())[:, 0].contiguous()
                    l_energy = torch.sum(pe.nll_rows(rcf, xf, rcf.to_rows(energy_norm.float())[:, 0].contiguous(), pe.cond_vec(g), nez) / zsum)
            # (A stream each for the pitch and the energy chain — forked from the encoder's stream, re-joined to it — was tried again in
            # round 3 with the capture-stream fix of train.Trainer in place: eager steps run, the capture still dies in
            # hipStreamEndCapture (host segmentation fault, 8 s into the run, before any replay).  Three streams in a capture work, a
            # fourth forked from a FORKED stream does not on this ROCm; not pursued further.)
        return l_length, l_pitch, l_energy, logw
