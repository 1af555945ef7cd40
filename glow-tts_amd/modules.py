"""Host-side mirror of the reference's modules.py for the hot path (same class names, constructor
arguments and state_dict keys), with the arithmetic in HIP kernels behind the C-ABI.

  ActNorm        reference modules.py:575-619
  InvConvNear    reference modules.py:622-668
  WN             reference modules.py:105-179
  LayerNorm      reference modules.py:26-44
Parameters are ordinary nn.Parameters (fp32 masters) owned by PyTorch; kernels borrow pointers.
"""
import math

import torch
from torch import nn

from . import _lib, flow_impl
from .ops import PackSlice, PackedConv, RowsCtx


class ConvP(nn.Module):
    """Parameters of a plain nn.Conv1d (state_dict keys `weight`, `bias`) + packed bf16 images."""
    weight_norm = False

    def __init__(self, in_channels, out_channels, kernel_size, gate=False, zero_init=False, split3=False):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.gate = in_channels, out_channels, kernel_size, gate
        self.split3 = split3                       # bf16x3 images (ops.PackedConv(split3=True)): near-fp32 1x1 GEMM
        w = torch.empty(out_channels, in_channels, kernel_size)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))                 # nn.Conv1d default init
        bound = 1 / math.sqrt(in_channels * kernel_size)
        b = torch.empty(out_channels).uniform_(-bound, bound)
        if zero_init:
            w.zero_(); b.zero_()
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(b)
        self._pc = None
        self.cat_slice = None                      # set by an owner that runs this conv as a window of a concatenated GEMM
        # a SECOND pair of images in MFMA-fragment order, for the fused between-WaveNets kernels (csrc/wn_boundary.hip);
        # the row-major pair stays (synthesis, the stand-alone module API and the reference path of the tests use it)
        self.also_frag = False
        self.pc_frag = None

    @property
    def pc(self):
        if self._pc is None:
            raise RuntimeError("conv weights not packed: call prepare() first")
        return self._pc

    def _new_pc(self, Cout=None):
        return PackedConv(Cout or self.out_channels, self.in_channels, self.kernel_size, self.gate, device=self.weight.device,
                          norm_only=self.cat_slice is not None, split3=self.split3)

    def _ensure_frag(self, dev):
        if self.also_frag and (self.pc_frag is None or self.pc_frag.inv_norm.device != dev):
            self.pc_frag = PackedConv(self.out_channels, self.in_channels, self.kernel_size, False, device=dev, frag=True)
        if not self.also_frag:
            self.pc_frag = None

    def _ensure_pcs(self):
        if self._pc is None or self._pc.inv_norm.device != self.weight.device:
            self._pc = self._new_pc()
        self._ensure_frag(self.weight.device)

    def _pack_entries(self):
        if self.cat_slice is not None:             # the images live in the owner's concatenated GEMM
            return [(self.weight, None, self.cat_slice())]
        return [(self.weight, None, self._pc)] + ([(self.weight, None, self.pc_frag)] if self.pc_frag is not None else [])

    def prepare(self):
        """(Re)pack the current weights for the MFMA kernels — once per optimizer step."""
        self._ensure_pcs()
        for v, g, pc in self._pack_entries():
            _pack_one(pc, v, g)
        return self


class WNConvP(ConvP):
    """Parameters of torch.nn.utils.weight_norm(nn.Conv1d): `weight_g` [Cout,1,1], `weight_v`, `bias`."""
    weight_norm = True

    def __init__(self, in_channels, out_channels, kernel_size, gate=False, split_res_skip=False):
        super().__init__(in_channels, out_channels, kernel_size, gate)
        v = self.weight.data
        del self.weight
        self.weight_g = nn.Parameter(v.reshape(out_channels, -1).norm(dim=1).reshape(out_channels, 1, 1).clone())
        self.weight_v = nn.Parameter(v.clone())
        self.split_res_skip = split_res_skip       # res_skip layer of a WN: rows [0,h) residual, [h,2h) skip
        self.skip_slice = None                     # set by the owning WN: where the skip rows are packed
        self.skip_slice_frag = None                # ... and where in the fragment-ordered twin of that image (boundary kernels)
        self.pc_res = None
        self.frag = False                          # set by the owning WN (fused layer kernels): fragment-ordered images

    def _ensure_pcs(self):
        dev = self.weight_v.device
        if self._pc is None or self._pc.inv_norm.device != dev:
            in_wn = self.skip_slice is not None
            self._pc = PackedConv(self.out_channels, self.in_channels, self.kernel_size, self.gate, device=dev, norm_only=in_wn,
                                  frag=self.frag and not in_wn, gate16=self.frag and self.gate)
            if self.split_res_skip:
                self.pc_res = PackedConv(self.out_channels // 2, self.in_channels, self.kernel_size, False, device=dev, frag=self.frag)
        self._ensure_frag(dev)

    def _pack_entries(self):
        out = [(self.weight_v, self.weight_g, self._pc)]                 # inv_norm (+ images unless norm_only)
        h = self.out_channels // 2 if self.split_res_skip else 0
        if self.split_res_skip:
            out.append((self.weight_v[:h], self.weight_g[:h], self.pc_res))
        if self.skip_slice is not None:
            out.append((self.weight_v[h:], self.weight_g[h:], self.skip_slice()))
            if self.skip_slice_frag is not None:
                out.append((self.weight_v[h:], self.weight_g[h:], self.skip_slice_frag()))
        elif self.pc_frag is not None:
            out.append((self.weight_v, self.weight_g, self.pc_frag))
        return out

    def prepare(self):
        self._ensure_pcs()
        for v, g, pc in self._pack_entries():
            _pack_one(pc, v, g)
        return self


def _pack_one(pc, v, g):
    """Single-conv packing (tests / tools; training packs everything in one launch through _PackPlan)."""
    L = _lib.lib()
    v = v.detach().contiguous().float()
    gg = None if g is None else g.detach().reshape(-1).contiguous().float()
    km = getattr(pc, "km", 1)
    _lib.check(L.gt_pack_conv_weights(_lib.ptr(v), _lib.ptr(gg), _lib.ptr(pc.fwd), _lib.ptr(pc.dgrad), _lib.ptr(pc.inv_norm),
                                      pc.Cout, pc.Cin, pc.taps, max(pc.Np_f, pc.Cout), max(pc.Kp_f, km * pc.Cin),
                                      max(pc.Np_d, pc.Cin), max(pc.Kp_d, km * pc.Cout), pc.flags,
                                      _lib.current_stream(v.device)), "gt_pack_conv_weights")


def _pack_key(entry):
    """identity of one pack job: where the parameter lives, where its image goes, in which format (PackSlice views are
    re-created per call, so object identity would not do)"""
    v, _, pc = entry
    return (v.data_ptr(), 0 if pc.fwd is None else pc.fwd.data_ptr(), 0 if pc.dgrad is None else pc.dgrad.data_ptr(), pc.flags)


class _PackPlan:
    """All conv weights of a module tree packed by ONE kernel launch (gt_pack_conv_weights_multi).
    The descriptor table lives on the device and is rebuilt only when a parameter's storage moves."""

    def __init__(self, module):
        import ctypes
        entries = []                                     # (v, g, PackedConv)
        for m in module.modules():
            if isinstance(m, ConvP):
                if getattr(m, "no_pack", False):
                    continue
                m._ensure_pcs()
                entries += m._pack_entries()
            elif hasattr(m, "_pack_entries_extra"):
                entries += m._pack_entries_extra()
        self.n = len(entries)
        self.key = tuple(_pack_key(e) for e in entries)
        self.keep = entries
        self.tables = []                                 # (device table, n, rows, group8)
        if self.n == 0:
            return
        dev = entries[0][0].device
        # the bf16x3 (split) images are written by the one-row-per-workgroup kernel: a table of their own
        def g8(pc, maxn):
            return pc.Cout % 8 == 0 and pc.Cin % 8 == 0 and pc.Cin * pc.taps <= maxn and not getattr(pc, "split3", False)

        plain = [e for e in entries if not getattr(e[2], "split3", False)]
        # rows of <= 1024 elements (the decoder's 192 x 5 convs: 90 % of the weights) go through the small form of the packing kernel
        # (16 KB of LDS per workgroup instead of 36: twice the resident workgroups), the rest through the 2304-element form
        small = [e for e in plain if g8(e[2], 1024)] if all(g8(e[2], 2304) for e in plain) else []
        large = [e for e in plain if not any(e is x for x in small)]
        for part in (small, large, [e for e in entries if getattr(e[2], "split3", False)]):
            if not part:
                continue
            arr = (_lib.PackDesc * len(part))()
            row = 0
            for d, (v, g, pc) in zip(arr, part):
                d.v, d.g = v.data_ptr(), (g.data_ptr() if g is not None else None)
                d.pack_fwd = pc.fwd.data_ptr() if pc.fwd is not None else None
                d.pack_dgrad = pc.dgrad.data_ptr() if pc.dgrad is not None else None
                d.inv_norm = pc.inv_norm.data_ptr() if pc.inv_norm is not None else None
                d.Cout, d.Cin, d.taps = pc.Cout, pc.Cin, pc.taps
                d.Np_fwd, d.Kp_fwd, d.Np_dgrad, d.Kp_dgrad, d.gate, d.row_start = pc.Np_f, pc.Kp_f, pc.Np_d, pc.Kp_d, pc.flags, row
                row += pc.Cout
            # 8 output channels per workgroup (coalesced 16-byte stores of both images) when every conv allows it
            group8 = 2 if part is small else int(all(g8(pc, 2304) for _, _, pc in part))
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.tables.append((raw.to(dev), len(part), row, group8))

    def run(self):
        if self.n == 0:
            return
        dev = self.keep[0][0].device
        for table, n, rows, group8 in self.tables:
            _lib.check(_lib.lib().gt_pack_conv_weights_multi(_lib.ptr(table), n, rows, group8, _lib.current_stream(dev)),
                       "gt_pack_conv_weights_multi")


_RETIRED_PLANS = []


def prepare_all(module, side=None, join=True):
    """(Re)pack every conv weight under `module` for the MFMA kernels — once per optimizer step.
    side: a stream that already follows the current one (train.Trainer: the encoder's branch, with the step's accumulator fills
    on it): the small gathers of the derived biases go there, beside the packing launch, and the current stream joins it at the end."""
    wns, mhas = [], []
    for m in module.modules():
        if isinstance(m, WN):
            wns.append(m)
        elif hasattr(m, "conv_q") and hasattr(m, "qkv_bias"):
            mhas.append(m)
        elif hasattr(m, "_refresh_padded"):
            m._refresh_padded()
    import contextlib
    with torch.no_grad(), (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        # derived biases of every WN (sum of the layers' skip biases) and every attention layer (q | k | v): ONE
        # gather + ONE reduction for the whole model instead of two small launches per module
        # (into PERSISTENT buffers: a captured step that leaves the packing to the trainer — FlowGenerator.external_prepare —
        # has their addresses baked into its kernel arguments)
        keep = module.__dict__.setdefault("_derived_bias", {})
        if wns and all(w.n_layers == wns[0].n_layers and w.hidden_channels == wns[0].hidden_channels for w in wns):
            H, n = wns[0].hidden_channels, wns[0].n_layers
            dev = wns[0].res_skip_layers[0].bias.device
            sb = keep.get("skip")
            if sb is None or sb.shape != (len(wns), H) or sb.device != dev:
                sb = keep["skip"] = torch.empty(len(wns), H, device=dev)
            torch.sum(torch.cat([rs.bias[-H:] for w in wns for rs in w.res_skip_layers]).view(len(wns), n, H), 1, out=sb)
            for w, row in zip(wns, sb):
                w.skip_bias = row
        else:
            for w in wns:
                w._refresh_padded()
        if mhas:
            C = mhas[0].channels
            dev = mhas[0].conv_q.bias.device
            qb = keep.get("qkv")
            if qb is None or qb.shape != (len(mhas), 3 * C) or qb.device != dev:
                qb = keep["qkv"] = torch.empty(len(mhas), 3 * C, device=dev)
            torch.cat([c.bias for a in mhas for c in (a.conv_q, a.conv_k, a.conv_v)], out=qb.view(-1))
            for a, row in zip(mhas, qb):
                a.qkv_bias = row
    plan = getattr(module, "_pack_plan", None)
    if plan is not None:
        cur = []
        for m in module.modules():                       # same walk as _PackPlan: parameter storage AND packed-image identity
            if isinstance(m, ConvP):
                if getattr(m, "no_pack", False):
                    continue
                m._ensure_pcs()
                cur += [_pack_key(e) for e in m._pack_entries()]
            elif hasattr(m, "_pack_entries_extra"):
                cur += [_pack_key(e) for e in m._pack_entries_extra()]
        if tuple(cur) != plan.key:
            _RETIRED_PLANS.append(plan)      # a captured graph has the old plan's descriptor table baked in: keep it allocated
            plan = None
    if plan is None:
        plan = _PackPlan(module)
        object.__setattr__(module, "_pack_plan", plan)
    plan.run()
    if side is not None and join:                 # join=False: the caller makes the current stream wait for `side` itself, later
        torch.cuda.current_stream().wait_stream(side)


class _RowsFn(torch.autograd.Function):
    """Generic bridge: runner.forward(*tensors) -> (outputs, saved); runner.backward(saved, *grads)
    -> grads aligned with the tensor inputs."""

    @staticmethod
    def forward(ctx, runner, n_out, *tensors):
        outs, saved = runner.forward(*tensors)
        ctx.set_materialize_grads(False)          # runners take None for an unused output's gradient (no zero fills)
        ctx.runner, ctx.saved, ctx.n_in = runner, saved, len(tensors)
        ctx.mark_non_differentiable(*[o for o in outs[n_out:]])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        gin = ctx.runner.backward(ctx.saved, *grads)
        assert len(gin) == ctx.n_in
        return (None, None) + tuple(gin)


def _mask_lengths(x_mask):
    return x_mask.sum([1, 2]).to(torch.int32)


class LayerNorm(nn.Module):
    """reference modules.LayerNorm (modules.py:26-44): over the channel dim, eps 1e-4."""

    def __init__(self, channels, eps=1e-4):
        super().__init__()
        self.channels, self.eps = channels, eps
        self.gamma = nn.Parameter(torch.ones(channels))
        self.beta = nn.Parameter(torch.zeros(channels))


class ActNorm(nn.Module):
    """reference modules.ActNorm (modules.py:575-619)."""

    def __init__(self, channels, ddi=False, **kwargs):
        super().__init__()
        self.channels = channels
        self.initialized = not ddi
        self.logs = nn.Parameter(torch.zeros(1, channels, 1))
        self.bias = nn.Parameter(torch.zeros(1, channels, 1))

    def store_inverse(self):
        pass

    def set_ddi(self, ddi):
        """modules.py:604-605: the next forward (of the owning FlowSpecDecoder) sets logs / bias from the masked batch
        statistics of this layer's input (ActNorm.initialize, modules.py:607-619 -> gt_actnorm_ddi), once."""
        self.initialized = not ddi


class InvConvNear(nn.Module):
    """reference modules.InvConvNear (modules.py:622-668): QR-orthogonal init with det > 0."""

    def __init__(self, channels, n_split=4, no_jacobian=False, **kwargs):
        super().__init__()
        assert n_split == 4, "the fused kernel implements n_split=4 (every reference config)"
        self.channels, self.n_split, self.no_jacobian = channels, n_split, no_jacobian
        w_init = torch.linalg.qr(torch.FloatTensor(n_split, n_split).normal_())[0]
        if torch.det(w_init) < 0:
            w_init[:, 0] = -1 * w_init[:, 0]
        self.weight = nn.Parameter(w_init)

    def store_inverse(self):
        """modules.py:667-668 (the kernels take W^-T from the cached flow scalars of FlowSpecDecoder.store_inverse)."""
        self.weight_inv = torch.inverse(self.weight.float()).to(dtype=self.weight.dtype)


class WN(nn.Module):
    """reference modules.WN (modules.py:105-179): gated conv stack, all convs weight-normed."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0):
        super().__init__()
        assert kernel_size % 2 == 1 and hidden_channels % 2 == 0
        assert dilation_rate == 1, "dilation_rate is 1 in every reference config (configs/*.json); kernels assume it"
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.kernel_size, self.dilation_rate, self.n_layers = kernel_size, dilation_rate, n_layers
        self.gin_channels, self.p_dropout = gin_channels, p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        if gin_channels != 0:
            self.cond_layer = WNConvP(gin_channels, 2 * hidden_channels * n_layers, 1)
        for i in range(n_layers):
            self.in_layers.append(WNConvP(hidden_channels, 2 * hidden_channels, kernel_size, gate=True))
            last = i == n_layers - 1
            self.res_skip_layers.append(WNConvP(hidden_channels, hidden_channels if last else 2 * hidden_channels, 1,
                                                split_res_skip=not last))
        # output = sum_i skip_i(acts_i) (modules.py:168-170) is ONE GEMM over the K-concatenated gated activations:
        # every layer's skip rows are packed into a window of pc_skipcat ([H, n_layers*H])
        self._pc_skipcat = None
        self._pc_skipcat_frag = None               # its fragment-ordered twin (set_boundary_frag: csrc/wn_boundary.hip reads it)
        self.skip_bias = None
        # one kernel per layer (csrc/wn_layer.hip) for the shape every reference config has; set_fused(False) gives round 1's
        # two-kernels-per-layer launch sequence (kept as the reference the fused kernels are tested against)
        self.fused = False
        self.set_fused(hidden_channels == 192 and kernel_size == 5)
        for i, rs in enumerate(self.res_skip_layers):
            rs.skip_slice = (lambda i=i: PackSlice(self.pc_skipcat, i * self.hidden_channels, self.hidden_channels, self.hidden_channels))

    def set_fused(self, on):
        """Switch between the fused layer kernels (fragment-ordered weight images) and the two-kernel path (row-major
        images); the packed images are rebuilt by the next prepare_all()."""
        on = bool(on) and self.hidden_channels == 192 and self.kernel_size == 5
        if on == self.fused and getattr(self, "_fused_set", False):
            return
        self.fused, self._fused_set = on, True
        for il in self.in_layers:
            il.frag, il._pc = on, None
        for rs in self.res_skip_layers:
            rs.frag, rs._pc, rs.pc_res = on, None, None

    def set_stack(self, fwd=True, bwd=True):
        """One launch per WaveNet and direction (csrc/wn_stack.hip) where that pays, or one launch per layer (csrc/wn_layer.hip: the
        path the stack kernels are tested against, and what long batches fall back to) — a property of THIS module (round 2 kept it
        in module-level switches that tests assigned)."""
        self.stack_fwd, self.stack_bwd = bool(fwd), bool(bwd)

    def set_boundary_frag(self, on):
        """Also keep the skip-cat GEMM's images in MFMA-fragment order (the fused between-WaveNets kernels)."""
        H = self.hidden_channels
        for i, rs in enumerate(self.res_skip_layers):
            rs.skip_slice_frag = (lambda i=i: PackSlice(self.pc_skipcat_frag, i * H, H, H)) if on else None
        if not on:
            self._pc_skipcat_frag = None

    @property
    def pc_skipcat_frag(self):
        dev = self.in_layers[0].weight_v.device
        if self._pc_skipcat_frag is None or self._pc_skipcat_frag.fwd.device != dev:
            self._pc_skipcat_frag = PackedConv(self.hidden_channels, self.n_layers * self.hidden_channels, 1, False, device=dev, frag=True)
        return self._pc_skipcat_frag

    @property
    def pc_skipcat(self):
        dev = self.in_layers[0].weight_v.device
        if self._pc_skipcat is None or self._pc_skipcat.fwd.device != dev:
            self._pc_skipcat = PackedConv(self.hidden_channels, self.n_layers * self.hidden_channels, 1, False, device=dev)
        return self._pc_skipcat

    def _refresh_padded(self):
        """Bias of the concatenated skip GEMM = sum of the layers' skip biases (once per optimizer step)."""
        H = self.hidden_channels
        with torch.no_grad():
            self.skip_bias = torch.stack([rs.bias[-H:] for rs in self.res_skip_layers]).sum(0)


class WNP(WN):
    """reference modules.WNP (modules.py:272-362): WN's gated conv stack with PER-FRAME conditioning — cond_layer1
    (weight-normed 1x1 conv, gin_channels1 -> 2*hidden*n_layers/n_sqz) is applied to the un-squeezed pitch / energy
    contour and squeezed like the mel (modules.py:353-362); the identity when the contour is None (modules.py:323-324).
    The loop itself is WN's (flow_impl.wn_fwd with cond_per_row)."""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels1=0, n_sqz=2):
        super().__init__(0, hidden_channels, kernel_size, dilation_rate, n_layers, 0, p_dropout)
        assert n_sqz == 2 and gin_channels1 in (0, 1), "the reference builds WNP(…, 1, n_sqz=2) (attentions.py:113-114)"
        self.n_sqz, self.gin_channels1 = n_sqz, gin_channels1
        if gin_channels1 != 0:
            self.cond_layer1 = WNConvP(gin_channels1, 2 * hidden_channels * n_layers // n_sqz, 1)
            self.cond_layer1.no_pack = True          # C_in = 1: an outer product, applied in the decoder runner


class ConvReluNorm(nn.Module):
    """reference modules.ConvReluNorm (modules.py:70-102): the text-encoder prenet."""

    def __init__(self, in_channels, hidden_channels, out_channels, kernel_size, n_layers, p_dropout):
        super().__init__()
        assert n_layers > 1, "Number of layers should be larger than 0."
        assert in_channels == hidden_channels == out_channels, "prenet kernels assume equal widths (models.py:674)"
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, hidden_channels, out_channels
        self.kernel_size, self.n_layers, self.p_dropout = kernel_size, n_layers, p_dropout
        self.conv_layers = nn.ModuleList([ConvP(in_channels if i == 0 else hidden_channels, hidden_channels, kernel_size)
                                          for i in range(n_layers)])
        self.norm_layers = nn.ModuleList([LayerNorm(hidden_channels) for _ in range(n_layers)])
        self.proj = ConvP(hidden_channels, out_channels, 1, zero_init=True)      # modules.py:91-92
