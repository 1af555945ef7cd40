"""Host-side mirror of the fork's stochastic predictors, arithmetic in HIP kernels behind the C-ABI
(csrc/predictor_ops.hip + the bf16 MFMA GEMM for the 192x192 1x1 convs).

  DilatedDepthSeparableConv   reference modules.py:683-735
  ElementwiseAffine           reference modules.py:738-756
  ConvFlow                    reference modules.py:759-819   (spline: transforms.py:12-202)
  StochasticDurationPredictor reference models.py:217-333
  StochasticPitchPredictor    reference models.py:335-407
  StochasticEnergyPredictor   reference models.py:409-481

Same class names, constructor arguments, forward signatures and state_dict keys as the reference.  Each predictor's
training forward (the negative log-likelihood) is ONE autograd node whose forward / backward are explicit kernel launch
sequences on the rows layout; the noise the reference draws with torch.randn inside forward can be injected (`noise=`),
which is how the parity tests pin the stochastic paths.
"""
import math

import torch
from torch import nn

from . import _lib, ops, wgrad
from .flow_impl import conv_param_grads
from .modules import ConvP, LayerNorm, _RowsFn, _mask_lengths, prepare_all
from .ops import RowsCtx, conv_rows, grad_accumulator

LN_EPS = 1e-5                     # modules.LayerNorm2 (modules.py:57)


def _st(dev):
    return _lib.current_stream(dev)


def _bias_grad_f32(conv, dy, grads):
    """bias gradient of a 1x1 conv from the fp32 output-gradient rows (column sums, gt_colsum): the bf16 copy that feeds
    the weight-gradient GEMM would cost the bias ~2^-8 per summand for nothing"""
    db = grad_accumulator(conv.bias)
    _lib.check(_lib.lib().gt_colsum(_lib.ptr(dy), dy.stride(0), 1, _lib.ptr(db), dy.shape[0], dy.shape[1], _st(dy.device)), "gt_colsum")
    grads[conv.bias] = db


def _split3_rows(x, rc=None):
    """fp32 (or bf16) rows [R, C] -> bf16x3 rows [R, 3C] = [hi | hi | lo]: the operand layout of the split (near-fp32) 1x1
    GEMMs.  The predictors' spline flows amplify bf16 rounding of their conditioning chaotically (relative L2
    errors > 1 on parameter gradients with plain bf16 operands, < 1e-2 with the split), so every 192x192 product here is
    x_hi w_hi + x_hi w_lo + x_lo w_hi on the bf16 MFMA GEMM — three passes of a tiny GEMM instead of an fp32 kernel."""
    R, C = x.shape
    out = torch.empty(R, 3 * C, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().gt_rows_split3(_lib.ptr(x), x.stride(0), int(x.dtype == torch.float32), _lib.ptr(out), 3 * C, None, R, C, _st(x.device)),
               "gt_rows_split3")
    return out


class DilatedDepthSeparableConv(nn.Module):
    def __init__(self, channels, kernel_size, num_layers, dropout_p=0.0):
        super().__init__()
        assert kernel_size == 3 and channels == 192, "kernels implement kernel_size 3 at 192 channels (every reference use)"
        self.channels, self.kernel_size, self.num_layers, self.dropout_p = channels, kernel_size, num_layers, dropout_p
        self.convs_sep = nn.ModuleList()
        self.convs_1x1 = nn.ModuleList()
        self.norms_1 = nn.ModuleList()
        self.norms_2 = nn.ModuleList()
        for _ in range(num_layers):
            sep = ConvP(1, channels, kernel_size)            # depthwise: weight [C, 1, k] (groups = C, modules.py:710-712)
            sep.no_pack = True
            self.convs_sep.append(sep)
            self.convs_1x1.append(ConvP(channels, channels, 1, split3=True))
            self.norms_1.append(LayerNorm(channels, eps=LN_EPS))
            self.norms_2.append(LayerNorm(channels, eps=LN_EPS))

    def forward(self, x, x_mask, g=None):
        """modules.py:718-735 as a stand-alone module: x [b, C, t], x_mask [b, 1, t], g [b, C, t] or None."""
        prepare_all(self)
        runner = _DDSRunner(self, x_mask, self.training, g is not None)
        (out,) = _RowsFn.apply(runner, 1, x, *([g] if g is not None else []), *runner.params)
        return out


def dds_fwd(rc, dds, x, train, seed, want_bf16=False):
    """x: fp32 rows [R, C] (masked; the `+ g` of modules.py:724-725 already applied).  -> (out fp32, out bf16 | None, saved)"""
    L = _lib.lib()
    dev = x.device
    R, C = x.shape
    utt = rc.row_utt()
    p = dds.dropout_p if train else 0.0
    saved = []
    outb = None
    for i in range(dds.num_layers):
        sep, n1, c1, n2 = dds.convs_sep[i], dds.norms_1[i], dds.convs_1x1[i], dds.norms_2[i]
        a1 = torch.empty(R, 3 * C, dtype=torch.bfloat16, device=dev)              # bf16x3: [hi | hi | lo]
        _lib.check(L.gt_dds_sep_fwd(_lib.ptr(x), x.stride(0), _lib.ptr(sep.weight), _lib.ptr(sep.bias), _lib.ptr(n1.gamma), _lib.ptr(n1.beta),
                                    _lib.ptr(utt), _lib.ptr(rc.rowmask), _lib.ptr(a1), 3 * C, R, C, dds.kernel_size ** i, LN_EPS, _st(dev)),
                   "gt_dds_sep_fwd")
        h2 = conv_rows(a1, c1.pc, rc, bias=c1.bias, out_f32=True)
        out = torch.empty(R, C, dtype=torch.float32, device=dev)
        last = i == dds.num_layers - 1
        _lib.check(L.gt_dds_out_fwd(_lib.ptr(h2), _lib.ptr(x), x.stride(0), _lib.ptr(n2.gamma), _lib.ptr(n2.beta), _lib.ptr(rc.rowmask),
                                    _lib.ptr(out), None, R, C, LN_EPS, float(p), int(seed + i),
                                    _lib.ptr(ops.seed_word(dev)) if p > 0 else None, _st(dev)), "gt_dds_out_fwd")
        saved.append((x, a1, h2))
        x = out
    if want_bf16:
        outb = _split3_rows(x)
    return x, outb, (saved, p, seed)


def dds_bwd(rc, dds, saved_all, dy, grads):
    """dy: fp32 rows, gradient at dds_fwd's output -> gradient at its input; parameter gradients into `grads`."""
    L = _lib.lib()
    saved, p, seed = saved_all
    dev = dy.device
    R, C = dy.shape
    utt = rc.row_utt()
    # inside a module's backward (an open WgradQueue) the three kernels leave their parameter gradients as per-workgroup partial rows
    # and ONE launch adds them up when the queue is flushed; each of them used to end in 384 - 768 same-address atomics per workgroup
    # (556 workgroups at cfg 5's 17.8 k frame rows: 32 - 65 us per launch, ~180 of them per step)
    q = wgrad.active()
    n_part = L.gt_dds_bwd_partial_rows(R) if q is not None else 0

    def part(width):
        return torch.empty(n_part, width, dtype=torch.float32, device=dev) if q is not None else None

    for i in reversed(range(dds.num_layers)):
        sep, n1, c1, n2 = dds.convs_sep[i], dds.norms_1[i], dds.convs_1x1[i], dds.norms_2[i]
        x, a1, h2 = saved[i]
        dg2, db2 = grad_accumulator(n2.gamma), grad_accumulator(n2.beta)
        dh2 = torch.empty(R, 3 * C, dtype=torch.bfloat16, device=dev)             # bf16x3
        pt = part(2 * C)
        _lib.check(L.gt_dds_out_bwd(_lib.ptr(h2), _lib.ptr(dy), _lib.ptr(n2.gamma), _lib.ptr(n2.beta), _lib.ptr(rc.rowmask), _lib.ptr(dh2),
                                    _lib.ptr(dg2), _lib.ptr(db2), _lib.ptr(pt), R, C, LN_EPS, float(p), int(seed + i),
                                    _lib.ptr(ops.seed_word(dev)) if p > 0 else None, _st(dev)), "gt_dds_out_bwd")
        if pt is not None:
            q.add_ln(pt, dg2, db2)
        grads[n2.gamma], grads[n2.beta] = dg2, db2
        grads.update(conv_param_grads(c1, a1[:, :C], dh2[:, :C], R))               # weight gradient from the hi parts
        da1 = conv_rows(dh2, c1.pc, rc, dgrad=True, out_f32=True)
        dg1, db1 = grad_accumulator(n1.gamma), grad_accumulator(n1.beta)
        dh1 = torch.empty(R, C, dtype=torch.float32, device=dev)
        _lib.check(L.gt_dds_sep_bwd(_lib.ptr(x), x.stride(0), _lib.ptr(sep.weight), _lib.ptr(sep.bias), _lib.ptr(n1.gamma), _lib.ptr(n1.beta),
                                    _lib.ptr(utt), _lib.ptr(rc.rowmask), _lib.ptr(da1), _lib.ptr(dh1), _lib.ptr(dg1), _lib.ptr(db1),
                                    _lib.ptr(pt1 := part(2 * C)), R, C, dds.kernel_size ** i, LN_EPS, _st(dev)), "gt_dds_sep_bwd")
        if pt1 is not None:
            q.add_ln(pt1, dg1, db1)
        grads[n1.gamma], grads[n1.beta] = dg1, db1
        dw, db = grad_accumulator(sep.weight), grad_accumulator(sep.bias)
        dx = torch.empty(R, C, dtype=torch.float32, device=dev)
        _lib.check(L.gt_dds_dw_bwd(_lib.ptr(x), x.stride(0), _lib.ptr(dh1), _lib.ptr(dy), _lib.ptr(sep.weight), _lib.ptr(utt),
                                   _lib.ptr(rc.rowmask), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(pt2 := part(4 * C)), R, C,
                                   dds.kernel_size ** i, _st(dev)), "gt_dds_dw_bwd")
        if pt2 is not None:
            q.add_ln(pt2, dw, db)
        grads[sep.weight], grads[sep.bias] = dw, db
        dy = dx
    return dy


class _DDSRunner:
    def __init__(self, dds, x_mask, train, has_g, seed=0):
        self.dds, self.x_mask, self.train, self.has_g, self.seed = dds, x_mask, train, has_g, seed
        self.params = list(dds.parameters())

    def forward(self, x, *rest):
        rc = RowsCtx(_mask_lengths(self.x_mask), x.shape[2])
        xin = x.detach().float() + (rest[0].detach().float() if self.has_g else 0)
        out, _, saved = dds_fwd(rc, self.dds, rc.to_rows(xin * self.x_mask), self.train, self.seed)
        return (rc.from_rows(out),), (rc, saved)

    def backward(self, saved_all, dout):
        rc, saved = saved_all
        grads = {}
        with wgrad.WgradQueue(dout.device, site=self.dds):
            dx = dds_bwd(rc, self.dds, saved, rc.to_rows(dout.float() * self.x_mask), grads)
        d = rc.from_rows(dx)
        return [d] + ([d] if self.has_g else []) + [grads.get(p) for p in self.params]


class ElementwiseAffine(nn.Module):
    def __init__(self, channels):
        super().__init__()
        assert channels == 2
        self.channels = channels
        self.translation = nn.Parameter(torch.zeros(channels, 1))
        self.log_scale = nn.Parameter(torch.zeros(channels, 1))


class ConvFlow(nn.Module):
    def __init__(self, in_channels, hidden_channels, kernel_size, num_layers, num_bins=10, tail_bound=5.0):
        super().__init__()
        assert in_channels == 2 and num_bins == 10 and tail_bound == 5.0, "the spline kernels implement the reference's configuration"
        self.num_bins, self.tail_bound, self.hidden_channels, self.half_channels = num_bins, tail_bound, hidden_channels, in_channels // 2
        self.pre = ConvP(self.half_channels, hidden_channels, 1)
        self.pre.no_pack = True                                       # one input channel: an outer product, fused with `+ g`
        self.convs = DilatedDepthSeparableConv(hidden_channels, kernel_size, num_layers, dropout_p=0.0)
        self.proj = ConvP(hidden_channels, self.half_channels * (num_bins * 3 - 1), 1, zero_init=True)     # modules.py:788-790
        self.proj.no_pack = True                                      # 29 rows, exact fp32, fused with the spline


def _ea_fwd(rc, ea, z, acc, sign=-1.0, reverse=False):
    out = torch.empty_like(z)
    _lib.check(_lib.lib().gt_ea_fwd(_lib.ptr(z), _lib.ptr(ea.log_scale), _lib.ptr(ea.translation), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()),
                                    _lib.ptr(out), _lib.ptr(acc), float(sign), int(reverse), z.shape[0], _st(z.device)), "gt_ea_fwd")
    return out


def _ea_bwd(rc, ea, z_in, dz, gacc, grads, sign=-1.0):
    dx = torch.empty_like(dz)
    dls, dtr = grad_accumulator(ea.log_scale), grad_accumulator(ea.translation)
    _lib.check(_lib.lib().gt_ea_bwd(_lib.ptr(z_in), _lib.ptr(ea.log_scale), _lib.ptr(dz), _lib.ptr(gacc), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()),
                                    _lib.ptr(dx), _lib.ptr(dls), _lib.ptr(dtr), float(sign), z_in.shape[0], _st(dz.device)), "gt_ea_bwd")
    grads[ea.log_scale], grads[ea.translation] = dls, dtr
    return dx


def _cf_fwd(rc, cf, z, g1, g2, acc, sign=-1.0):
    """ConvFlow forward + the channel flip that follows it (models.py:314-318).  z [R,2] fp32 rows."""
    L = _lib.lib()
    dev = z.device
    R = z.shape[0]
    C = cf.hidden_channels
    x0 = torch.empty(R, C, dtype=torch.float32, device=dev)
    _lib.check(L.gt_convflow_pre_fwd(_lib.ptr(z), 2, _lib.ptr(cf.pre.weight), _lib.ptr(cf.pre.bias), _lib.ptr(g1), _lib.ptr(g2),
                                     _lib.ptr(rc.rowmask), _lib.ptr(x0), R, C, _st(dev)), "gt_convflow_pre_fwd")
    h, _, sv = dds_fwd(rc, cf.convs, x0, False, 0)
    zo = torch.empty_like(z)
    par = torch.empty(R, 32, dtype=torch.float32, device=dev)
    _lib.check(L.gt_convflow_spline_fwd(_lib.ptr(h), _lib.ptr(cf.proj.weight), _lib.ptr(cf.proj.bias), _lib.ptr(z), _lib.ptr(rc.rowmask),
                                        _lib.ptr(rc.row_utt()), _lib.ptr(zo), _lib.ptr(par), _lib.ptr(acc), float(sign), 1, R, C, _st(dev)),
               "gt_convflow_spline_fwd")
    return zo, (z, sv, h, par)


def _cf_bwd(rc, cf, saved, dzo, gacc, dg, grads, sign=-1.0):
    """-> dz_in; dg [R,C] (+)= gradient at the conditioning rows (None: not wanted)"""
    L = _lib.lib()
    z, sv, h, par = saved
    dev = z.device
    R = z.shape[0]
    C = cf.hidden_channels
    dh = torch.empty(R, C, dtype=torch.float32, device=dev)
    dz = torch.empty_like(z)
    dWp, dbp = grad_accumulator(cf.proj.weight), grad_accumulator(cf.proj.bias)
    q = wgrad.active()                               # inside a module's backward: partial rows + the queue's one reduce launch
    pt = torch.empty(L.gt_convflow_spline_partial_rows(R), L.gt_convflow_spline_partial_width(), dtype=torch.float32, device=dev) \
        if (q is not None and dWp.numel() + dbp.numel() == L.gt_convflow_spline_partial_width()) else None
    _lib.check(L.gt_convflow_spline_bwd(_lib.ptr(h), _lib.ptr(cf.proj.weight), _lib.ptr(par), _lib.ptr(z), _lib.ptr(dzo), _lib.ptr(gacc),
                                        _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()), _lib.ptr(dh), _lib.ptr(dWp), _lib.ptr(dbp), _lib.ptr(pt),
                                        _lib.ptr(dz), float(sign), 1, R, C, _st(dev)), "gt_convflow_spline_bwd")
    if pt is not None:
        q.add_ln(pt, dWp, dbp)
    grads[cf.proj.weight], grads[cf.proj.bias] = dWp, dbp
    dx0 = dds_bwd(rc, cf.convs, sv, dh, grads)
    dwp, dbpre = grad_accumulator(cf.pre.weight), grad_accumulator(cf.pre.bias)
    _lib.check(L.gt_convflow_pre_bwd(_lib.ptr(dx0), _lib.ptr(z), 2, _lib.ptr(cf.pre.weight), _lib.ptr(rc.rowmask), _lib.ptr(dwp), _lib.ptr(dbpre),
                                     _lib.ptr(dz), 2, _lib.ptr(dg), R, C, _st(dev)), "gt_convflow_pre_bwd")
    grads[cf.pre.weight], grads[cf.pre.bias] = dwp, dbpre
    return dz


def _cf_rev(rc, cf, z, g1):
    """ConvFlow.forward(reverse=True) (modules.py:805-819): z [R,2] -> [z0, RQS^-1(z1)] (no flip here)."""
    L = _lib.lib()
    dev = z.device
    R, C = z.shape[0], cf.hidden_channels
    x0 = torch.empty(R, C, dtype=torch.float32, device=dev)
    _lib.check(L.gt_convflow_pre_fwd(_lib.ptr(z), 2, _lib.ptr(cf.pre.weight), _lib.ptr(cf.pre.bias), _lib.ptr(g1), None,
                                     _lib.ptr(rc.rowmask), _lib.ptr(x0), R, C, _st(dev)), "gt_convflow_pre_fwd")
    h, _, _ = dds_fwd(rc, cf.convs, x0, False, 0)
    zo = torch.empty_like(z)
    _lib.check(L.gt_convflow_spline_inv(_lib.ptr(h), _lib.ptr(cf.proj.weight), _lib.ptr(cf.proj.bias), _lib.ptr(z), _lib.ptr(rc.rowmask),
                                        _lib.ptr(zo), R, C, _st(dev)), "gt_convflow_spline_inv")
    return zo


def flows_fwd(rc, flows, z, g1, g2, acc):
    """[ElementwiseAffine, ConvFlow x n] with a flip after every ConvFlow; acc[utt] -= log|det| of every flow."""
    saved = []
    for idx, f in enumerate(flows):
        if idx == 0:
            saved.append(z)
            z = _ea_fwd(rc, f, z, acc)
        else:
            z, sv = _cf_fwd(rc, f, z, g1, g2, acc)
            saved.append(sv)
    return z, saved


def flows_bwd(rc, flows, saved, dz, gacc, dg, grads):
    for idx in reversed(range(len(flows))):
        if idx == 0:
            dz = _ea_bwd(rc, flows[0], saved[0], dz, gacc, grads)
        else:
            dz = _cf_bwd(rc, flows[idx], saved[idx], dz, gacc, dg, grads)
    return dz


def _nll_gauss(rc, z, acc):
    _lib.check(_lib.lib().gt_nll_gauss_fwd(_lib.ptr(z), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()), _lib.ptr(acc), z.shape[0], _st(z.device)),
               "gt_nll_gauss_fwd")


def _nll_gauss_bwd(rc, z, gacc):
    dz = torch.empty_like(z)
    _lib.check(_lib.lib().gt_nll_gauss_bwd(_lib.ptr(z), _lib.ptr(gacc), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()), _lib.ptr(dz), z.shape[0],
                                           _st(z.device)), "gt_nll_gauss_bwd")
    return dz


class _PredictorBase(nn.Module):
    """What the three predictors share (models.py:232-235, 348-351, 422-425 and the `cond` convs): the text-side condition
    encoder pre -> (+ cond(g) [+ cond_lang(l)]) -> convs -> proj, and the flow stack `flows`."""

    def _build(self, in_channels, kernel_size, p_dropout, n_flows, gin_channels, lin_channels=0):
        C = in_channels                                     # filter_channels = in_channels (models.py:223, 339, 413)
        self.in_channels, self.filter_channels, self.kernel_size, self.p_dropout, self.n_flows = C, C, kernel_size, p_dropout, n_flows
        self.gin_channels, self.lin_channels = gin_channels, lin_channels
        self.pre = ConvP(C, C, 1, split3=True)
        self.convs = DilatedDepthSeparableConv(C, kernel_size, num_layers=3, dropout_p=p_dropout)
        self.proj = ConvP(C, C, 1, split3=True)
        self.flows = nn.ModuleList([ElementwiseAffine(2)] + [ConvFlow(2, C, kernel_size, num_layers=3) for _ in range(n_flows)])
        self._step = 0

    def cond_vec(self, g, l=None):
        """cond(detach(g)) (+ cond_lang(detach(l))) for g [b, gin, 1], l [b, lin, 1] -> [b, C] (B rows: host-side plumbing,
        differentiable w.r.t. the cond parameters); the pre GEMM's epilogue adds it to every row of its utterance."""
        F = torch.nn.functional
        v = None
        if g is not None:
            v = F.linear(g.detach().squeeze(-1), self.cond.weight.squeeze(-1), self.cond.bias)
        if l is not None:
            vl = F.linear(l.detach().squeeze(-1), self.cond_lang.weight.squeeze(-1), self.cond_lang.bias)
            v = vl if v is None else v + vl
        return v

    def _kernel_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("cond.") and not n.startswith("cond_lang.")]

    # ---- shared pieces on rows ----------------------------------------------------------------------------------------
    def _cond_fwd(self, rc, xb, vec, train, seed):
        """xb: bf16 rows of the (detached) text-side features -> conditioning rows fp32 [R, C] (masked) + saved"""
        xb3 = _split3_rows(xb)
        x0 = conv_rows(xb3, self.pre.pc, rc, bias=self.pre.bias, cond=None if vec is None else vec.detach().float().contiguous(),
                       mask=True, out_f32=True)
        x1, x1b, sv = dds_fwd(rc, self.convs, x0, train, seed, want_bf16=True)
        xc = conv_rows(x1b, self.proj.pc, rc, bias=self.proj.bias, mask=True, out_f32=True)
        return xc, (xb, sv, x1b)

    def _cond_bwd(self, rc, saved, dxc, grads, want_dvec):
        xb, sv, x1b = saved
        R, C = dxc.shape
        dxcb = _split3_rows(dxc)
        grads.update(conv_param_grads(self.proj, x1b[:, :C], dxcb[:, :C], R, want_bias=False))
        _bias_grad_f32(self.proj, dxc, grads)
        dx1 = conv_rows(dxcb, self.proj.pc, rc, dgrad=True, out_f32=True, mask=True)
        dx0 = dds_bwd(rc, self.convs, sv, dx1, grads)
        dx0b = _split3_rows(dx0)
        grads.update(conv_param_grads(self.pre, xb, dx0b[:, :C], R, want_bias=False))
        _bias_grad_f32(self.pre, dx0, grads)
        return ops.cond_grad(rc, dx0) if want_dvec else None

    def _reverse_rows(self, rc, xb, vec, noise_rows):
        """reverse=True branch (models.py:324-333, 398-407): flows reversed, the useless vflow dropped, z flipped before
        every flow; returns channel 0 as rows [R]."""
        xc, _ = self._cond_fwd(rc, xb, vec, False, 0)
        order = list(reversed(range(len(self.flows))))
        order = order[:-2] + [order[-1]]
        z = (noise_rows * rc.rowmask[:, None]).contiguous()
        for idx in order:
            z = z.flip(1).contiguous()
            z = _ea_fwd(rc, self.flows[0], z, None, reverse=True) if idx == 0 else _cf_rev(rc, self.flows[idx], z, xc)
        return z[:, 0].contiguous()

    def _rows_io(self, x, x_mask):
        rc = RowsCtx(_mask_lengths(x_mask), x.shape[2])
        return rc, rc.to_rows(x.detach().float() * x_mask, torch.bfloat16)


class _NllRunner:
    """nll [B] of a predictor as one autograd node over its parameters (+ the per-utterance cond vector)."""

    def __init__(self, pred, rc, xb, dr, noise, train, seed, has_vec):
        self.pred, self.rc, self.xb, self.dr, self.noise, self.train, self.seed, self.has_vec = pred, rc, xb, dr, noise, train, seed, has_vec
        self.params = pred._kernel_params()

    def forward(self, *rest):
        vec = rest[0] if self.has_vec else None
        nll, saved = self.pred._nll_fwd(self.rc, self.xb, self.dr, self.noise, vec, self.train, self.seed)
        return (nll,), saved

    def backward(self, saved, gout):
        grads = {}
        dev = self.rc.device
        gacc = gout.detach().float().contiguous()
        with wgrad.WgradQueue(dev, site=self.pred):
            dvec = self.pred._nll_bwd(self.rc, saved, gacc, grads, self.has_vec)
        return ([dvec] if self.has_vec else []) + [grads.get(p) for p in self.params]


class StochasticDurationPredictor(_PredictorBase):
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, n_flows=4, gin_channels=0, lin_channels=0, emoin_channels=0):
        super().__init__()
        assert emoin_channels == 0, "cond_emo is never used by the reference's forward (models.py:269-271)"
        self._build(in_channels, kernel_size, p_dropout, n_flows, gin_channels, lin_channels)
        C = self.filter_channels
        self.post_pre = ConvP(1, C, 1)
        self.post_pre.no_pack = True
        self.post_convs = DilatedDepthSeparableConv(C, kernel_size, num_layers=3, dropout_p=p_dropout)
        self.post_proj = ConvP(C, C, 1, split3=True)
        self.post_flows = nn.ModuleList([ElementwiseAffine(2)] + [ConvFlow(2, C, kernel_size, num_layers=3) for _ in range(n_flows)])
        if gin_channels != 0:
            self.cond = nn.Conv1d(gin_channels, C, 1)
        if lin_channels != 0:
            self.cond_lang = nn.Conv1d(lin_channels, C, 1)

    def _nll_fwd(self, rc, xb, w, noise, vec, train, seed):
        """xb bf16 rows [R,C], w fp32 rows [R] (durations), noise fp32 rows [R,2] -> nll [B] (models.py:280-322)"""
        L = _lib.lib()
        dev = xb.device
        R, C = xb.shape[0], self.filter_channels
        xc, s_c = self._cond_fwd(rc, xb, vec, train, seed)
        hw0 = torch.empty(R, C, dtype=torch.float32, device=dev)
        _lib.check(L.gt_convflow_pre_fwd(_lib.ptr(w), 1, _lib.ptr(self.post_pre.weight), _lib.ptr(self.post_pre.bias), None, None,
                                         _lib.ptr(rc.rowmask), _lib.ptr(hw0), R, C, _st(dev)), "gt_convflow_pre_fwd")
        hw1, hw1b, s_h = dds_fwd(rc, self.post_convs, hw0, train, seed + 8, want_bf16=True)
        h = conv_rows(hw1b, self.post_proj.pc, rc, bias=self.post_proj.bias, mask=True, out_f32=True)
        acc = torch.zeros(rc.B, dtype=torch.float32, device=dev)
        e_q = (noise * rc.rowmask[:, None]).contiguous()
        z_q, s_q = flows_fwd(rc, self.post_flows, e_q, xc, h, acc)
        z = torch.empty_like(z_q)
        _lib.check(L.gt_sdp_mid_fwd(_lib.ptr(z_q), _lib.ptr(w), _lib.ptr(e_q), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()), _lib.ptr(z),
                                    _lib.ptr(acc), R, _st(dev)), "gt_sdp_mid_fwd")
        z_f, s_f = flows_fwd(rc, self.flows, z, xc, None, acc)
        _nll_gauss(rc, z_f, acc)
        return acc, (s_c, s_h, hw1b, s_q, z_q, w, s_f, z_f)

    def _nll_bwd(self, rc, saved, gacc, grads, want_dvec):
        L = _lib.lib()
        s_c, s_h, hw1b, s_q, z_q, w, s_f, z_f = saved
        dev = z_f.device
        R, C = z_f.shape[0], self.filter_channels
        dxc = torch.zeros(R, C, dtype=torch.float32, device=dev)
        dz = _nll_gauss_bwd(rc, z_f, gacc)
        dz = flows_bwd(rc, self.flows, s_f, dz, gacc, dxc, grads)
        dzq = torch.empty_like(dz)
        _lib.check(L.gt_sdp_mid_bwd(_lib.ptr(z_q), _lib.ptr(w), _lib.ptr(dz), _lib.ptr(gacc), _lib.ptr(rc.rowmask), _lib.ptr(rc.row_utt()),
                                    _lib.ptr(dzq), R, _st(dev)), "gt_sdp_mid_bwd")
        dsum = torch.zeros(R, C, dtype=torch.float32, device=dev)                 # gradient at (xc + h), the posterior's condition
        flows_bwd(rc, self.post_flows, s_q, dzq, gacc, dsum, grads)
        dxc += dsum
        # h = post_proj(post_convs(post_pre(w))) * mask
        dhb = _split3_rows(dsum)
        grads.update(conv_param_grads(self.post_proj, hw1b[:, :C], dhb[:, :C], R, want_bias=False))
        _bias_grad_f32(self.post_proj, dsum, grads)
        dhw1 = conv_rows(dhb, self.post_proj.pc, rc, dgrad=True, out_f32=True, mask=True)
        dhw0 = dds_bwd(rc, self.post_convs, s_h, dhw1, grads)
        dwp, dbp = grad_accumulator(self.post_pre.weight), grad_accumulator(self.post_pre.bias)
        _lib.check(L.gt_convflow_pre_bwd(_lib.ptr(dhw0), _lib.ptr(w), 1, _lib.ptr(self.post_pre.weight), _lib.ptr(rc.rowmask), _lib.ptr(dwp),
                                         _lib.ptr(dbp), None, 0, None, R, C, _st(dev)), "gt_convflow_pre_bwd")
        grads[self.post_pre.weight], grads[self.post_pre.bias] = dwp, dbp
        return self._cond_bwd(rc, s_c, dxc, grads, want_dvec)

    def nll_rows(self, rc, xb, w_rows, vec, noise_rows=None):
        """training branch on rows (FlowGenerator's own call): -> nll [B]"""
        self._step += 1
        if noise_rows is None:
            noise_rows = torch.randn(rc.R, 2, dtype=torch.float32, device=xb.device)            # models.py:288
        runner = _NllRunner(self, rc, xb, w_rows, noise_rows, self.training, (self._step * 40503) & 0x7fffffff, vec is not None)
        (nll,) = _RowsFn.apply(runner, 1, *([vec] if vec is not None else []), *runner.params)
        return nll

    def forward(self, x, x_mask, dr=None, g=None, l=None, emo=None, reverse=False, noise_scale=1.0, noise=None):
        """models.py:261-333.  x [b,C,t] (detached inside), x_mask [b,1,t], dr [b,1,t] durations; noise [b,2,t] replaces the
        torch.randn draw (reverse: before the noise_scale factor)."""
        prepare_all(self)
        rc, xb = self._rows_io(x, x_mask)
        vec = self.cond_vec(g, l)
        b, _, t = x.shape
        nz = noise if noise is not None else torch.randn(b, 2, t, device=x.device, dtype=torch.float32)
        nrows = rc.to_rows(nz.float())
        if reverse:
            with torch.no_grad():
                out = self._reverse_rows(rc, xb, vec, nrows * noise_scale)
            return rc.from_rows(out[:, None].contiguous())
        assert dr is not None
        return self.nll_rows(rc, xb, rc.to_rows(dr.float() * x_mask)[:, 0].contiguous(), vec, nrows)


class StochasticPitchPredictor(_PredictorBase):
    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, n_flows=4, gin_channels=0, emoin_channels=0):
        super().__init__()
        self._build(in_channels, kernel_size, p_dropout, n_flows, gin_channels)
        self.emoin_channels = emoin_channels
        if gin_channels != 0:
            self.cond = nn.Conv1d(gin_channels, self.filter_channels, 1)
        if emoin_channels != 0:
            self.cond_emo = nn.Conv1d(emoin_channels, self.filter_channels, 1)     # never used by forward (models.py:372-374)

    def _kernel_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("cond")]

    def _nll_fwd(self, rc, xb, dr, noise, vec, train, seed):
        """z = cat(dr, noise * mask) -> flows -> nll [B] (models.py:379-396)"""
        dev = xb.device
        xc, s_c = self._cond_fwd(rc, xb, vec, train, seed)
        z = (torch.stack([dr, noise], dim=1) * rc.rowmask[:, None]).contiguous()
        acc = torch.zeros(rc.B, dtype=torch.float32, device=dev)
        z_f, s_f = flows_fwd(rc, self.flows, z, xc, None, acc)
        _nll_gauss(rc, z_f, acc)
        return acc, (s_c, s_f, z_f)

    def _nll_bwd(self, rc, saved, gacc, grads, want_dvec):
        s_c, s_f, z_f = saved
        dxc = torch.zeros(z_f.shape[0], self.filter_channels, dtype=torch.float32, device=z_f.device)
        dz = _nll_gauss_bwd(rc, z_f, gacc)
        flows_bwd(rc, self.flows, s_f, dz, gacc, dxc, grads)
        return self._cond_bwd(rc, s_c, dxc, grads, want_dvec)

    def nll_rows(self, rc, xb, dr_rows, vec, noise_rows=None):
        self._step += 1
        if noise_rows is None:
            noise_rows = torch.randn(rc.R, dtype=torch.float32, device=xb.device)               # models.py:383
        runner = _NllRunner(self, rc, xb, dr_rows, noise_rows, self.training, (self._step * 48271) & 0x7fffffff, vec is not None)
        (nll,) = _RowsFn.apply(runner, 1, *([vec] if vec is not None else []), *runner.params)
        return nll

    def forward(self, x, x_mask, dr=None, g=None, emo=None, reverse=False, noise_scale=1.0, noise=None):
        """models.py:364-407 (and :438-481 for the energy twin).  noise: [b,1,t] in training, [b,2,t] in reverse."""
        prepare_all(self)
        rc, xb = self._rows_io(x, x_mask)
        vec = self.cond_vec(g)
        b, _, t = x.shape
        if reverse:
            nz = noise if noise is not None else torch.randn(b, 2, t, device=x.device, dtype=torch.float32)
            with torch.no_grad():
                out = self._reverse_rows(rc, xb, vec, rc.to_rows(nz.float()) * noise_scale)
            return rc.from_rows(out[:, None].contiguous())
        assert dr is not None
        nz = noise if noise is not None else torch.randn(b, 1, t, device=x.device, dtype=torch.float32)
        return self.nll_rows(rc, xb, rc.to_rows(dr.float() * x_mask)[:, 0].contiguous(), vec, rc.to_rows(nz.float())[:, 0].contiguous())


class StochasticEnergyPredictor(StochasticPitchPredictor):
    """reference models.py:409-481: the pitch predictor's twin (argument order of the constructor differs)."""

    def __init__(self, in_channels, filter_channels, kernel_size, p_dropout, n_flows=4, emoin_channels=0, gin_channels=0):
        super().__init__(in_channels, filter_channels, kernel_size, p_dropout, n_flows, gin_channels=gin_channels, emoin_channels=emoin_channels)
