"""Host-side mirror of the reference's models.py for the hot path.

  FlowSpecDecoder   reference models.py:719-789  (squeeze, 12 x [ActNorm, InvConvNear, CouplingBlock], unsqueeze)
The forward signature, return values and state_dict keys (`flows.{3k}.logs`, `flows.{3k+1}.weight`,
`flows.{3k+2}.start.weight_v`, ...) are the reference's; the whole block chain runs as ONE autograd
node whose forward/backward are explicit HIP kernel launch sequences (flow_impl.py).
"""
import os

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

    def store_inverse(self):
        for f in self.flows:
            f.store_inverse()

    def forward(self, x, x_mask, g=None, emo=None, pitch=None, energy=None, reverse=False, prepared=False):
        """x: [b, 80, t] (t even after the caller's preprocess, models.py:1248-1253; an odd trailing
        frame is dropped like commons.squeeze does), x_mask: [b, 1, t].  Returns (z, logdet_tot)."""
        if pitch is not None or energy is not None:
            raise NotImplementedError("pitch/energy conditioning (cfg 5) is out of the round-1 scope")
        if not prepared:
            prepare_all(self)
        if reverse:                                  # inference direction (models.py:769-770,781-782): no log-det, no autograd
            with torch.no_grad():
                conds = _wn_cond_all([self.flows[3 * b + 2].wn for b in range(self.n_blocks)], g)
                return _DecoderRunner(self, x_mask, g is not None, False, 0).reverse(x.detach(), conds), None
        self._step += 1
        runner = _DecoderRunner(self, x_mask, g is not None, self.training, seed=(self._step * 7919) & 0x7fffffff)
        conds = []
        if g is not None:
            conds = _wn_cond_all([self.flows[3 * b + 2].wn for b in range(self.n_blocks)], g)
        z, logdet = _RowsFn.apply(runner, 2, x, *conds, *runner.params)
        return z, logdet


class _DecoderRunner:
    def __init__(self, dec, x_mask, has_cond, train, seed):
        self.dec, self.has_cond, self.train, self.seed = dec, has_cond, train, seed
        self.x_mask = x_mask
        self.params = [p for n, p in dec.named_parameters() if ".wn.cond_layer." not in n]

    def forward(self, x, *rest):
        L = _lib.lib()
        dec = self.dec
        nb = dec.n_blocks
        conds = rest[:nb] if self.has_cond else [None] * nb
        B, C, T = x.shape
        dev = x.device
        T2 = T // 2
        len_sq = (_mask_lengths(self.x_mask) // 2).to(torch.int32)          # mask[:, :, 1::2] (commons.py:348)
        rc = ops.make_ctx(len_sq, T2, "y", div=2)
        xin = x.detach().float().contiguous()
        rows = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        st = _lib.current_stream(dev)
        _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(xin), _lib.ptr(rows), _lib.ptr(rc.lengths), B, C, T, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        logdet = ops.zeros_small(B, torch.float32, dev)
        saved = []
        cur = rows
        for b in range(nb):
            an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
            y1, x0, s1 = flow_impl.actnorm_invconv_fwd(rc, cur, an.logs, an.bias, ic.weight, logdet)
            cur, s2 = flow_impl.coupling_fwd(rc, cb, y1, x0, conds[b], logdet, self.train, self.seed + 16 * b)
            saved.append((s1, s2))
        z = torch.empty(B, C, T2 * 2, dtype=torch.float32, device=dev)
        _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(z), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
        return (z.to(x.dtype), logdet), (rc, saved, (B, C, T))

    def reverse(self, z, conds):
        """z [B, C, T] -> x: flows in reverse order (coupling^-1, InvConvNear^-1, ActNorm^-1 per block)."""
        L = _lib.lib()
        dec = self.dec
        B, C, T = z.shape
        dev = z.device
        T2 = T // 2
        len_sq = (_mask_lengths(self.x_mask) // 2).to(torch.int32)
        rc = ops.make_ctx(len_sq, T2, "y", div=2)
        zin = z.float().contiguous()
        cur = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        st = _lib.current_stream(dev)
        _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(zin), _lib.ptr(cur), _lib.ptr(rc.lengths), B, C, T, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        x0 = torch.empty(rc.R, C, dtype=torch.bfloat16, device=dev)
        _lib.check(L.gt_rows_f32_to_bf16(_lib.ptr(cur), 2 * C, _lib.ptr(x0), C, None, rc.R, C, st), "gt_rows_f32_to_bf16")
        for b in reversed(range(dec.n_blocks)):
            an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
            cur = flow_impl.coupling_rev(rc, cb, cur, x0, conds[b])
            cur, x0 = flow_impl.actnorm_invconv_rev(rc, cur, an.logs, an.bias, ic.weight, want_x0=b > 0)
        x = torch.empty(B, C, T2 * 2, dtype=torch.float32, device=dev)
        _lib.check(L.gt_unsqueeze_rows_f32(_lib.ptr(cur), _lib.ptr(x), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_unsqueeze_rows_f32")
        return x.to(z.dtype)

    def backward(self, saved_all, dz, dlogdet):
        L = _lib.lib()
        rc, saved, (B, C, T) = saved_all
        dec = self.dec
        nb = dec.n_blocks
        dev = rc.device
        st = _lib.current_stream(dev)
        T2 = T // 2
        dlogdet = ops.zeros_small(B, torch.float32, dev) if dlogdet is None else dlogdet.contiguous().float()
        grads = {}
        drows = torch.empty(rc.R, 2 * C, dtype=torch.float32, device=dev)
        if dz is None:
            drows.zero_()
        else:
            dzc = dz.float().contiguous()
            _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(dzc), _lib.ptr(drows), _lib.ptr(rc.lengths), B, C, T2 * 2, rc.Tp, _lib.ptr(rc.row0), st), "gt_squeeze_rows_f32")
        dconds = [None] * nb
        cur = drows
        # data-gradient chain now; the weight gradients of every `chunk` blocks go out as one batch (wgrad.ASYNC: on a
        # side stream, beside the rest of the chain)
        chunk = int(os.environ.get("GT_WGRAD_CHUNK", "12"))
        for b1 in range(nb, 0, -chunk):
            b0 = max(0, b1 - chunk)
            with wgrad.WgradQueue(dev, site=dec.flows[3 * b0 + 2] if chunk < nb else dec):
                for b in reversed(range(b0, b1)):
                    an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
                    s1, s2 = saved[b]
                    cur, g2, dconds[b] = flow_impl.coupling_bwd(rc, cb, s2, cur, dlogdet, self.has_cond)
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
        return out + [grads.get(p) for p in self.params]


from .text_models import DurationPredictor, FlowGenerator, TextEncoder, mle_loss  # noqa: E402,F401
