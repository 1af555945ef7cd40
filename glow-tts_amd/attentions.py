"""Host-side mirror of the reference's attentions.py for the hot path (same names, constructor
arguments and state_dict keys); arithmetic in HIP kernels behind the C-ABI.

  CouplingBlock         reference attentions.py:89-194
  MultiHeadAttention    reference attentions.py:197-347
  FFN / Encoder         reference attentions.py:350-372 / 12-86
"""
import torch
from torch import nn

from . import flow_impl
from .modules import WN, ConvP, WNConvP, _RowsFn, _mask_lengths, prepare_all
from .ops import RowsCtx


def _wn_cond(wn, g):
    """cond_layer(g) (modules.py:148-149): a [B,gin]x[gin,2*H*n] product on B rows — host-side
    PyTorch plumbing (B <= 128 rows; differentiable w.r.t. g and the cond_layer parameters)."""
    if g is None:
        return None
    cl = wn.cond_layer
    v = cl.weight_v
    w = v * (cl.weight_g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1))
    return torch.nn.functional.conv1d(g, w, cl.bias).squeeze(-1).contiguous()


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
        if with_prosody_wn:
            raise NotImplementedError("wn_pitch / wn_energy (cfg 5) are outside the round-1 hot-path scope")

    def store_inverse(self):
        pass

    def forward(self, x, x_mask=None, reverse=False, g=None, emo=None, pitch=None, energy=None, **kwargs):
        if reverse:
            raise NotImplementedError("reverse flow (inference) is out of the training hot-path scope")
        if pitch is not None or energy is not None:
            raise NotImplementedError("pitch/energy conditioning (cfg 5) is out of the round-1 scope")
        prepare_all(self)
        runner = _CouplingRunner(self, x_mask, g is not None, self.training)
        tensors = [x] + ([_wn_cond(self.wn, g)] if g is not None else []) + runner.params
        z, logdet = _RowsFn.apply(runner, 2, *tensors)
        return z, logdet


class _CouplingRunner:
    def __init__(self, cb, x_mask, has_cond, train, seed=0):
        self.cb, self.has_cond, self.train, self.seed = cb, has_cond, train, seed
        self.x_mask = x_mask
        self.params = [p for n, p in cb.named_parameters() if not n.startswith("wn.cond_layer")]

    def forward(self, x, *rest):
        cond = rest[0] if self.has_cond else None
        B, C, T = x.shape
        rc = RowsCtx(_mask_lengths(self.x_mask), T)
        xr = rc.to_rows(x.detach().float() * self.x_mask)
        x0 = xr[:, :C // 2].to(torch.bfloat16)
        logdet = torch.zeros(B, dtype=torch.float32, device=x.device)
        z, saved = flow_impl.coupling_fwd(rc, self.cb, xr, x0, cond, logdet, self.train, self.seed)
        return (rc.from_rows(z), logdet), (rc, saved)

    def backward(self, saved_all, dz, dlogdet):
        rc, saved = saved_all
        dlogdet = torch.zeros(rc.B, device=dz.device) if dlogdet is None else dlogdet.contiguous().float()
        dzr = rc.to_rows(dz.float())
        dx, grads, dcond = flow_impl.coupling_bwd(rc, self.cb, saved, dzr, dlogdet, self.has_cond)
        out = [rc.from_rows(dx)]
        if self.has_cond:
            out.append(dcond)
        return out + [grads.get(p) for p in self.params]
