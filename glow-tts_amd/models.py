"""Host-side mirror of the reference's models.py for the hot path.

  FlowSpecDecoder   reference models.py:719-789  (squeeze, 12 x [ActNorm, InvConvNear, CouplingBlock], unsqueeze)
The forward signature, return values and state_dict keys (`flows.{3k}.logs`, `flows.{3k+1}.weight`,
`flows.{3k+2}.start.weight_v`, ...) are the reference's; the whole block chain runs as ONE autograd
node whose forward/backward are explicit HIP kernel launch sequences (flow_impl.py).
"""
import torch
from torch import nn

from . import _lib, flow_impl, wgrad
from .attentions import CouplingBlock, _wn_cond_all
from .modules import ActNorm, InvConvNear, _RowsFn, _mask_lengths, prepare_all
from . import ops
from .ops import HALO, RowsCtx


class FlowSpecDecoder(nn.Module):
    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_blocks, n_layers, p_dropout=0.,
                 n_split=4, n_sqz=2, sigmoid_scale=False, gin_channels=0, emoin_channels=0, with_prosody_wn=False):
        super().__init__()
        assert n_sqz == 2 and n_split == 4, "kernels implement n_sqz=2, n_split=4 (every reference config)"
        self.in_channels, self.hidden_channels, self.kernel_size = in_channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_blocks, self.n_layers, self.p_dropout = dilation_rate, n_blocks, n_layers, p_dropout
        self.n_split, self.n_sqz, self.sigmoid_scale, self.gin_channels = n_split, n_sqz, sigmoid_scale, gin_channels
        self.flows = nn.ModuleList()
        for b in range(n_blocks):
            self.flows.append(ActNorm(channels=in_channels * n_sqz))
            self.flows.append(InvConvNear(channels=in_channels * n_sqz, n_split=n_split))
            self.flows.append(CouplingBlock(in_channels * n_sqz, hidden_channels, kernel_size=kernel_size,
                                            dilation_rate=dilation_rate, n_layers=n_layers, gin_channels=gin_channels,
                                            emoin_channels=emoin_channels, p_dropout=p_dropout,
                                            sigmoid_scale=sigmoid_scale, n_sqz=n_sqz, with_prosody_wn=with_prosody_wn))
        self._step = 0
        self._inv_cache = None
        self.fused_boundary = False
        self.set_fused_boundary(True)

    def set_fused_boundary(self, on):
        """One kernel for everything between two WaveNets (skip GEMM, end conv, coupling, ActNorm, InvConvNear, start conv;
        csrc/wn_boundary.hip) when every block has the shape the kernel is built for; off = round 1's launch sequence
        (the reference path of the tests).  Takes effect with the next prepare_all()."""
        oks = [self.flows[3 * b + 2].set_boundary_fused(on) for b in range(self.n_blocks)]
        self.fused_boundary = bool(on) and all(oks)
        if not self.fused_boundary:
            for b in range(self.n_blocks):
                self.flows[3 * b + 2].set_boundary_fused(False)
        return self.fused_boundary

    def store_inverse(self):
        """models.py:787-789 -> modules.py:667-668 / attentions.py:188-194: freeze the decoder for synthesis.  The
        reference caches W^-1 and folds the weight norms away; here the packed bf16 weight images and every block's flow
        scalars (sum logs, logdet W, W^-T) are computed ONCE and reverse calls reuse them (no per-call re-pack) until
        `clear_inverse()` or a training-mode forward."""
        for f in self.flows:
            f.store_inverse()
        prepare_all(self)
        with torch.no_grad():
            self._inv_cache = [flow_impl.flow_scalars(self.flows[3 * b].logs, self.flows[3 * b + 1].weight)
                               for b in range(self.n_blocks)]

    def clear_inverse(self):
        self._inv_cache = None

    def forward(self, x, x_mask, g=None, emo=None, pitch=None, energy=None, reverse=False, prepared=False):
        """x: [b, 80, t] (t even after the caller's preprocess, models.py:1248-1253; an odd trailing
        frame is dropped like commons.squeeze does), x_mask: [b, 1, t].  Returns (z, logdet_tot).
        pitch / energy: [b, 1, t] (or [b, t], attentions.py:137-141) contours at the un-squeezed frame rate, or None:
        per-frame conditioning of every block's wn_pitch / wn_energy (cfg 5; needs with_prosody_wn=True)."""
        if (pitch is not None or energy is not None) and not hasattr(self.flows[2], "wn_pitch"):
            raise ValueError("pitch / energy conditioning needs FlowSpecDecoder(with_prosody_wn=True)")
        pitch = pitch.unsqueeze(1) if (pitch is not None and pitch.dim() == 2) else pitch
        energy = energy.unsqueeze(1) if (energy is not None and energy.dim() == 2) else energy
        cached = reverse and self._inv_cache is not None
        if not prepared and not cached:
            prepare_all(self)
        if not reverse:
            self._inv_cache = None               # parameters may move: the synthesis cache is stale
        wns = [self.flows[3 * b + 2].wn for b in range(self.n_blocks)]
        if reverse:                                  # inference direction (models.py:769-770,781-782): no log-det, no autograd
            with torch.no_grad():
                conds = _wn_cond_all(wns, g)
                runner = _DecoderRunner(self, x_mask, g is not None, False, 0, energy, pitch)
                return runner.reverse(x.detach(), conds, self._prosody_affine("wn_energy", energy), self._prosody_affine("wn_pitch", pitch)), None
        self._step += 1
        seed = (self._step * 7919) & 0x7fffffff
        conds = list(_wn_cond_all(wns, g)) if g is not None else []
        affs = [a for a in (self._prosody_affine("wn_energy", energy), self._prosody_affine("wn_pitch", pitch)) if a is not None]
        runner = _DecoderRunner(self, x_mask, g is not None, self.training, seed, energy, pitch)
        z, logdet = _RowsFn.apply(runner, 2, x, *conds, *affs, *runner.params)
        return z, logdet

    def _prosody_affine(self, which, contour):
        """cond_layer1 of every block's wn_energy / wn_pitch as (effective weight, bias) pairs [n_blocks, 2, 2*H*n/n_sqz]:
        with one input channel the weight-normed 1x1 conv (modules.py:289-291,320) is the per-frame affine map
        w[c] * contour + b[c]; the runner applies it to the squeezed contour.  Differentiable w.r.t. the parameters."""
        if contour is None:
            return None
        cls = [getattr(self.flows[3 * b + 2], which).cond_layer1 for b in range(self.n_blocks)]
        v = torch.stack([c.weight_v.reshape(c.out_channels, -1) for c in cls])        # [nb, O, 1]
        gg = torch.stack([c.weight_g.reshape(-1) for c in cls])                       # [nb, O]
        w = (v * (gg / v.norm(dim=2)).unsqueeze(-1)).squeeze(-1)
        return torch.stack([w, torch.stack([c.bias for c in cls])], dim=1)


class _DecoderRunner:
    def __init__(self, dec, x_mask, has_cond, train, seed, energy=None, pitch=None):
        self.dec, self.has_cond, self.train, self.seed = dec, has_cond, train, seed
        self.x_mask = x_mask
        self.cfg = getattr(dec, "rows_cfg", None) or ops.DEFAULT_ROWS        # the owning FlowGenerator's RowsConfig
        self.energy, self.pitch = energy, pitch                                   # [b,1,t] contours (no gradient) or None
        self.params = [p for n, p in dec.named_parameters() if ".wn.cond_layer." not in n and "cond_layer1" not in n]

    _contour_rows = staticmethod(lambda rc, c, B, T: flow_impl.contour_rows(rc, c, B, T))
    _cond_rows = staticmethod(flow_impl.cond_rows)

    def _split_inputs(self, rest):
        nb = self.dec.n_blocks
        k = nb if self.has_cond else 0
        conds = rest[:k] if self.has_cond else [None] * nb
        eaff = paff = None
        if self.energy is not None:
            eaff = rest[k]; k += 1
        if self.pitch is not None:
            paff = rest[k]; k += 1
        return conds, eaff, paff

    def forward(self, x, *rest):
        L = _lib.lib()
        dec = self.dec
        nb = dec.n_blocks
        conds, eaff, paff = self._split_inputs(rest)
        B, C, T = x.shape
        dev = x.device
        T2 = T // 2
        # squeezed lengths = mask[:, :, 1::2] (commons.py:348); a callable: a prebuilt (captured-graph) context has them already
        rc = ops.make_ctx(lambda: (_mask_lengths(self.x_mask) // 2).to(torch.int32), T2, "y", div=2, cfg=self.cfg)
        xin = x.detach().float().contiguous()
        st = _lib.current_stream(dev)
        logdet = ops.zeros_small(B, torch.float32, dev)
        esig, psig = self._contour_rows(rc, self.energy, B, T2 * 2), self._contour_rows(rc, self.pitch, B, T2 * 2)
        saved = []
        # one kernel between consecutive WaveNets (csrc/wn_boundary.hip) unless a block still waits for its data-dependent
        # init (that forward runs round 1's launch sequence once).  With per-frame prosody conditioning (cfg 5) a block is a chain
        # of up to three WaveNets: the boundary kernel sits behind the last one, a skip GEMM between two of them
        fused = dec.fused_boundary and all(dec.flows[3 * b].initialized for b in range(nb))
        pros = dict(esig=esig, eaff=eaff, psig=psig, paff=paff)
        # ... and then commons.squeeze / unsqueeze ride in its first / last launch (ragged rows, even T)
        folded = fused and T == T2 * 2 and getattr(rc, "rowbatch", None) is not None
        if folded:
            z = ops.zeros_big((B, C, T), torch.float32, dev)
            _, blocks = flow_impl.decoder_fwd_fused(rc, dec, None, conds, logdet, self.train, self.seed, y_bct=xin, z_bct=z, **pros)
            return (z.to(x.dtype), logdet), (rc, ("fused", blocks), (B, C, T), esig, psig)
        rows = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(xin), _lib.ptr(rows), _lib.ptr(rc.lengths), B, C, T, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        cur = rows
        if fused:
            cur, blocks = flow_impl.decoder_fwd_fused(rc, dec, rows, conds, logdet, self.train, self.seed, **pros)
            saved = ("fused", blocks)
        for b in range(0 if not fused else nb, nb):
            an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
            if not an.initialized:                                    # data-dependent init, once (modules.py:588-590)
                assert not torch.cuda.is_current_stream_capturing(), "run the DDI batch before capturing the step"
                flow_impl.actnorm_ddi(rc, cur, an)
            y1, x0, s1 = flow_impl.actnorm_invconv_fwd(rc, cur, an.logs, an.bias, ic.weight, logdet)
            cur, s2 = flow_impl.coupling_fwd(rc, cb, y1, x0, conds[b], logdet, self.train, self.seed + 16 * b,
                                             econd=self._cond_rows(esig, None if eaff is None else eaff[b]),
                                             pcond=self._cond_rows(psig, None if paff is None else paff[b]))
            saved.append((s1, s2))
        z = torch.empty(B, C, T2 * 2, dtype=torch.float32, device=dev)
        _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(z), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
        return (z.to(x.dtype), logdet), (rc, saved, (B, C, T), esig, psig)

    def reverse(self, z, conds, eaff=None, paff=None):
        """z [B, C, T] -> x: flows in reverse order (coupling^-1, InvConvNear^-1, ActNorm^-1 per block)."""
        L = _lib.lib()
        dec = self.dec
        B, C, T = z.shape
        dev = z.device
        T2 = T // 2
        len_sq = (_mask_lengths(self.x_mask) // 2).to(torch.int32)
        rc = ops.make_ctx(len_sq, T2, "y", div=2, cfg=self.cfg)
        zin = z.float().contiguous()
        cur = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        st = _lib.current_stream(dev)
        _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(zin), _lib.ptr(cur), _lib.ptr(rc.lengths), B, C, T, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        x0 = torch.empty(rc.R, C, dtype=torch.bfloat16, device=dev)
        _lib.check(L.gt_rows_f32_to_bf16(_lib.ptr(cur), 2 * C, _lib.ptr(x0), C, None, rc.R, C, st), "gt_rows_f32_to_bf16")
        esig, psig = self._contour_rows(rc, self.energy, B, T2 * 2), self._contour_rows(rc, self.pitch, B, T2 * 2)
        for b in reversed(range(dec.n_blocks)):
            an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
            cur = flow_impl.coupling_rev(rc, cb, cur, x0, conds[b], econd=self._cond_rows(esig, None if eaff is None else eaff[b]),
                                         pcond=self._cond_rows(psig, None if paff is None else paff[b]))
            cur, x0 = flow_impl.actnorm_invconv_rev(rc, cur, an.logs, an.bias, ic.weight, want_x0=b > 0,
                                                    scal=None if dec._inv_cache is None else dec._inv_cache[b])
        x = torch.empty(B, C, T2 * 2, dtype=torch.float32, device=dev)
        _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(x), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
        return x.to(z.dtype)

    def backward(self, saved_all, dz, dlogdet):
        L = _lib.lib()
        rc, saved, (B, C, T), esig, psig = saved_all
        dec = self.dec
        nb = dec.n_blocks
        dev = rc.device
        st = _lib.current_stream(dev)
        T2 = T // 2
        dlogdet = ops.zeros_small(B, torch.float32, dev) if dlogdet is None else dlogdet.contiguous().float()
        grads = {}
        fused = isinstance(saved, tuple) and len(saved) == 2 and saved[0] == "fused"
        O = 2 * dec.hidden_channels * dec.n_layers // 2
        deaff = torch.zeros(nb, 2, O, dtype=torch.float32, device=dev) if esig is not None else None     # (w, b) gradients of the
        dpaff = torch.zeros(nb, 2, O, dtype=torch.float32, device=dev) if psig is not None else None     # affine conditioning, per block
        if fused and dz is not None and T == T2 * 2 and getattr(rc, "rowbatch", None) is not None:
            # squeeze of d z / unsqueeze of the input gradient inside the first / last launch of the backward chain
            dzc = dz.float().contiguous()
            dx = ops.zeros_big((B, C, T), torch.float32, dev)
            ops.mark("dec bwd begin")
            with wgrad.WgradQueue(dev, site=dec):
                _, grads, dconds = flow_impl.decoder_bwd_fused(rc, dec, saved[1], None, dlogdet, self.has_cond, dz_bct=dzc, dx_bct=dx,
                                                               deaff=deaff, dpaff=dpaff)
                ops.mark("dec dgrad end")
            ops.mark("dec wgrad end")
            return [dx] + (dconds if self.has_cond else []) + [d for d in (deaff, dpaff) if d is not None] + [grads.get(p) for p in self.params]
        drows = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        if dz is None:
            drows.zero_()
        else:
            dzc = dz.float().contiguous()
            _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(dzc), _lib.ptr(drows), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        dconds = [None] * nb
        cur = drows
        # data-gradient chain now; ALL weight gradients of the decoder go out as one batch when the block ends
        with wgrad.WgradQueue(dev, site=dec):
            if fused:
                cur, grads, dconds = flow_impl.decoder_bwd_fused(rc, dec, saved[1], drows, dlogdet, self.has_cond, deaff=deaff, dpaff=dpaff)
            for b in (() if fused else reversed(range(nb))):
                an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
                s1, s2 = saved[b]
                if esig is None and psig is None:
                    cur, g2, dconds[b] = flow_impl.coupling_bwd(rc, cb, s2, cur, dlogdet, self.has_cond)
                else:
                    cur, g2, dconds[b], dpros = flow_impl.coupling_bwd(rc, cb, s2, cur, dlogdet, self.has_cond,
                                                                       econd=esig is not None, pcond=psig is not None)
                    # cond = w * contour + b per (frame, parity, channel): d w = sum dcond * contour, d b = sum dcond
                    for dc, sig, dst in ((dpros[0], esig, deaff), (dpros[1], psig, dpaff)):
                        if dc is not None:
                            d3 = dc.view(rc.R, 2, O)
                            dst[b, 0] = (d3 * sig[:, :, None]).sum((0, 1))
                            dst[b, 1] = d3.sum((0, 1))
                grads.update(g2)
                cur, g1 = flow_impl.actnorm_invconv_bwd(rc, s1, cur, dlogdet, an.logs, an.bias, ic.weight)
                grads.update(g1)
        dx = torch.zeros(B, C, T, dtype=torch.float32, device=dev) if T != T2 * 2 else torch.empty(B, C, T, dtype=torch.float32, device=dev)
        if T == T2 * 2:
            _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(dx), _lib.ptr(rc.lengths), B, C, T, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
        else:
            tmp = torch.empty(B, C, T2 * 2, dtype=torch.float32, device=dev)
            _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(tmp), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
            dx[:, :, :T2 * 2] = tmp
        out = [dx]
        if self.has_cond:
            out += dconds
        out += [d for d in (deaff, dpaff) if d is not None]
        return out + [grads.get(p) for p in self.params]


from .text_models import DurationPredictor, FlowGenerator, TextEncoder, mle_loss  # noqa: E402,F401
