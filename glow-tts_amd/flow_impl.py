"""Forward/backward launch sequences of the flow decoder on the rows layout (host side of the
HIP kernels).  Every function here only allocates buffers and calls the C-ABI; the hand-written
backward mirrors what autograd derives for the reference modules:

  actnorm_invconv_*   modules.ActNorm + modules.InvConvNear   (modules.py:584-599, 635-665)
  wn_*                modules.WN                               (modules.py:144-171)
  coupling_*          attentions.CouplingBlock                 (attentions.py:132-186)
"""
import torch

from . import _lib
from .ops import KERNEL_TIMER, RowsCtx, conv_rows, grad_accumulator, seed_word, zeros_small  # noqa: F401

_SCRATCH = {}
_SCRATCH_KEEP = []


def _scratch(name, nbytes, device):
    """Grow-only device scratch (wgrad partial slabs etc.), one per purpose and device.  A captured graph replays into the buffer it
    was captured with: the capture holds a reference to it (wgrad.capture_keep), so an outgrown buffer lives as long as those graphs."""
    from . import wgrad
    key = (name, str(device))
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _SCRATCH[key] = buf
    if buf.is_cuda and torch.cuda.is_current_stream_capturing() and not wgrad.capture_keep(buf):
        _SCRATCH_KEEP.append(buf)              # a capture nobody owns: for the life of the process
    return buf


def _st(dev):
    return _lib.current_stream(dev)


# ----------------------------------------------------------------------------- conv parameter grads
def conv_param_grads(conv, x, dy, R, want_bias=True, parts=None):
    """Gradients of a ConvP/WNConvP module's parameters from its input rows x (bf16) and output
    gradient rows dy (bf16): wgrad MFMA kernel -> slab partials -> weight-norm backward; bias by
    column sums.  Returns {param: grad}.  Inside a `wgrad.WgradQueue` block the work is only recorded
    (x and dy must then stay untouched until the block ends) and the returned tensors are filled by the
    queue's batched launches.  parts = [(dy_view, co_begin, co_count), ...] lets the output gradient live
    in several buffers (res/skip halves of modules.py:166-168); it needs an open queue."""
    from . import wgrad
    q = wgrad.active()
    if q is not None:
        pl = [(x, dy, 0, conv.pc.Cout)] if parts is None else [(x, d, b, c) for d, b, c in parts]
        return q.add(conv, R, pl, want_bias)
    if parts is not None:                     # immediate mode: gather the pieces into one [R, Cout] buffer
        dy = torch.cat([d[:, :c] for d, _, c in sorted(parts, key=lambda t: t[1])], dim=1)
    L = _lib.lib()
    pc = conv.pc
    dev = x.device
    import ctypes
    S = ctypes.c_int(0)
    nbytes = L.gt_conv_wgrad_workspace_bytes(R, pc.Cin, pc.Cout, pc.taps, ctypes.byref(S))
    ws = _scratch("wgrad", nbytes, dev)
    _lib.check(L.gt_conv_wgrad_bf16(_lib.ptr(x), x.stride(0), _lib.ptr(dy), dy.stride(0), R, pc.Cin, pc.Cout, pc.taps,
                                    _lib.ptr(ws), nbytes, _st(dev)), "gt_conv_wgrad_bf16")
    out = {}
    v = conv.weight_v if conv.weight_norm else conv.weight
    dv = torch.empty_like(v)
    db = torch.empty_like(conv.bias) if (want_bias and conv.bias is not None) else None
    if conv.weight_norm:
        dg = torch.empty_like(conv.weight_g)
        _lib.check(L.gt_weightnorm_bwd(_lib.ptr(ws), R, _lib.ptr(v), _lib.ptr(conv.weight_g), _lib.ptr(pc.inv_norm),
                                       _lib.ptr(dv), _lib.ptr(dg), _lib.ptr(db), pc.Cout, pc.Cin, pc.taps, 0, _st(dev)), "gt_weightnorm_bwd")
        out[conv.weight_v] = dv
        out[conv.weight_g] = dg
    else:
        _lib.check(L.gt_weightnorm_bwd(_lib.ptr(ws), R, _lib.ptr(v), None, None, _lib.ptr(dv), None, _lib.ptr(db),
                                       pc.Cout, pc.Cin, pc.taps, 0, _st(dev)), "gt_weightnorm_bwd")
        out[conv.weight] = dv
    if db is not None:
        out[conv.bias] = db
    return out


# ----------------------------------------------------------------------------- ActNorm + InvConvNear
def actnorm_ddi(rc, x, an):
    """ActNorm.initialize (modules.py:607-619) from the rows this layer is about to see; writes an.logs / an.bias."""
    L = _lib.lib()
    R, C = x.shape
    ws = torch.empty(2 * C, dtype=torch.float64, device=x.device)
    with torch.no_grad():
        _lib.check(L.gt_actnorm_ddi(_lib.ptr(x), _lib.ptr(rc.lengths), rc.B, R, C, _lib.ptr(ws), _lib.ptr(an.logs.data),
                                    _lib.ptr(an.bias.data), _st(x.device)), "gt_actnorm_ddi")
    an.initialized = True


def flow_scalars(logs, W):
    """scal[18] = {sum logs, logdet W, W^-T} (gt_flow_scalars) for one ActNorm / InvConvNear pair."""
    L = _lib.lib()
    lg = logs.detach().reshape(-1).contiguous()
    Wc = W.detach().contiguous()
    scal = torch.empty(18, dtype=torch.float32, device=lg.device)
    _lib.check(L.gt_flow_scalars(_lib.ptr(lg), lg.numel(), _lib.ptr(Wc), _lib.ptr(scal), _st(lg.device)), "gt_flow_scalars")
    return scal


def actnorm_invconv_fwd(rc, x, logs, bias, W, logdet, want_x0=True):
    """x: [R,C] fp32 rows.  Returns y [R,C] fp32, x0 bf16 [R,C/2] (coupling start input), saved."""
    L = _lib.lib()
    dev = x.device
    R, C = x.shape
    y = torch.empty_like(x)
    x0 = torch.empty(R, C // 2, dtype=torch.bfloat16, device=dev) if want_x0 else None
    scal = torch.empty(18, dtype=torch.float32, device=dev)
    lg = logs.detach().reshape(-1).contiguous()
    bs = bias.detach().reshape(-1).contiguous()
    Wc = W.detach().contiguous()
    _lib.check(L.gt_flow_scalars(_lib.ptr(lg), C, _lib.ptr(Wc), _lib.ptr(scal), _st(dev)), "gt_flow_scalars")
    _lib.check(L.gt_actnorm_invconv_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(x0), C // 2, _lib.ptr(lg), _lib.ptr(bs), _lib.ptr(Wc),
                                        _lib.ptr(scal), _lib.ptr(rc.rowmask), _lib.ptr(rc.lengths), _lib.ptr(logdet),
                                        rc.B, R, C, _st(dev)), "gt_actnorm_invconv_fwd")
    return y, x0, (x, lg, bs, Wc, scal)


def actnorm_invconv_bwd(rc, saved, dy, dlogdet, logs, bias, W):
    L = _lib.lib()
    x, lg, bs, Wc, scal = saved
    dev = x.device
    R, C = x.shape
    dx = torch.empty_like(x)
    dlogs = grad_accumulator(logs, (C,))
    dbias = grad_accumulator(bias, (C,))
    dW = grad_accumulator(W, (16,))
    _lib.check(L.gt_actnorm_invconv_bwd(_lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), _lib.ptr(lg), _lib.ptr(bs), _lib.ptr(Wc),
                                        _lib.ptr(scal), _lib.ptr(rc.rowmask), _lib.ptr(rc.lengths), _lib.ptr(dlogdet),
                                        _lib.ptr(dlogs), _lib.ptr(dbias), _lib.ptr(dW), rc.B, R, C, _st(dev)),
               "gt_actnorm_invconv_bwd")
    return dx, {logs: dlogs.view_as(logs), bias: dbias.view_as(bias), W: dW.view_as(W)}


# ----------------------------------------------------------------------------- WN
def _fused_ok(wn):
    """the one-kernel-per-layer path (csrc/wn_layer.hip): H = 192, k = 5, packed images exactly N rows high"""
    if not getattr(wn, "fused", False):
        return False
    pc = wn.in_layers[0].pc
    assert pc.frag and pc.gate16 and pc.Np_f == 2 * wn.hidden_channels and pc.Kp_f == wn.hidden_channels and \
        pc.Np_d == wn.hidden_channels and pc.Kp_d == 2 * wn.hidden_channels, "fused WN: fragment-ordered images expected (prepare_all after set_fused)"
    return True


_CUS = {}                    # device -> compute units (a property of the hardware, read once)


def _stack_pays(R, n, dev):
    """One launch per WaveNet (csrc/wn_stack.hip) owns 64 - 4 (n - 1) rows per workgroup where the per-layer kernels own 64: more
    workgroups for the same rows.  That is free while they fit the CUs at once (cfg 2: 188 instead of 152 of 256) and loses when it
    costs an extra round of workgroups (cfg 3's longest batches: 268 instead of 218)."""
    key = str(dev)
    if key not in _CUS:
        _CUS[key] = torch.cuda.get_device_properties(dev).multi_processor_count
    cus = _CUS[key]
    own = 64 - 4 * (n - 1)
    rounds = lambda rows: -(-(-(-R // rows)) // cus)
    return rounds(own) <= rounds(64)


def wn_fwd(rc, wn, h0, cond, train, seed, cond_per_row=False, layers_only=False, affine=None):
    """modules.WN.forward on rows.  h0: [R,H] bf16 (masked).  cond: [B, 2*H*n_layers] fp32 or None; with cond_per_row
    it is [R, 2*H*n_layers] — the per-frame conditioning of modules.WNP.forward (modules.py:316-343), whose loop is WN's.
    Returns out [R,H] bf16 (= skip sum * mask) and saved activations.

    One kernel per layer (gt_wn_layer_fwd: k=5 conv + gate + residual 1x1 on the gated tile).  The gated activations of
    all layers live side by side in ONE [R, n*H] buffer, and output = sum_i skip_i(acts_i) (modules.py:168-170) is a
    single K = n*H GEMM at the end instead of n read-modify-write passes over an fp32 accumulator.
    affine = (sig [R, 2], w [O], b [O]): modules.WNP's per-frame conditioning in its affine form (cond_layer1 has one input channel):
    the whole-WaveNet kernel forms it from the row's two contour values (no [R, 2*H*n] fp32 rows); the per-layer path
    materialises it (cond_rows) as before."""
    L = _lib.lib()
    R, H = h0.shape
    dev = h0.device
    n = wn.n_layers
    p = wn.p_dropout if train else 0.0
    if affine is not None:
        assert cond is None
        aff_sig, aff_w, aff_b = affine[0], affine[1].detach().float().contiguous(), affine[2].detach().float().contiguous()
    xs, ts, ss = [h0], [], []
    acts_all = torch.empty(R, n * H, dtype=torch.bfloat16, device=dev)
    x = h0
    fused = _fused_ok(wn)
    stamps = getattr(rc, "stamps", None)              # bench.py: live in-graph timing of the dominant kernel (ops.KernelStamps)
    if fused and getattr(wn, "stack_fwd", True) and n <= 4 and _stack_pays(R, n, dev):
        # all layers in ONE launch (csrc/wn_stack.hip: the 2-row halo between layers is recomputed, not exchanged)
        import ctypes
        ts = [torch.empty(R, H, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        ss = [torch.empty(R, H, dtype=torch.bfloat16, device=dev) for _ in range(n)]
        xs = [h0] + [torch.empty(R, H, dtype=torch.bfloat16, device=dev) for _ in range(n - 1)]
        pad = [None] * (4 - n)
        args = _lib.fill_args(
            _lib.WnStackFwdArgs, x0=h0, w_in=[il.pc.fwd for il in wn.in_layers] + pad, b_in=[il.bias for il in wn.in_layers] + pad,
            w_res=[rs.pc_res.fwd for rs in wn.res_skip_layers[:n - 1]] + [None] + pad,
            b_res=[rs.bias for rs in wn.res_skip_layers[:n - 1]] + [None] + pad,
            cond=cond, ldc=0 if cond is None else cond.stride(0), row0=rc.row0 if (cond is not None and not cond_per_row) else None,
            B=0 if (cond_per_row or cond is None) else rc.B, Tp=rc.Tp, rowmask=rc.rowmask, acts=acts_all, ldacts=acts_all.stride(0),
            gate_t=ts + pad, gate_s=ss + pad, x_out=xs[1:] + [None] + pad, R=R, H=H, taps=wn.kernel_size, n_layers=n,
            drop_p=float(p), drop_seed=int(seed), seed_dev=seed_word(dev) if p > 0 else None,
            stamps=stamps.buf if stamps else None, stamp_slot=stamps.take(f"stack{n}") if stamps else 0, stamp_base=stamps.base if stamps else None,
            **(dict(aff_w=aff_w, aff_b=aff_b, aff_sig=aff_sig) if affine is not None else {}))
        _ev = KERNEL_TIMER.start("wn_stack_fwd")
        rcode = L.gt_wn_stack_fwd(ctypes.byref(args), _st(dev))
        KERNEL_TIMER.stop(_ev)
        _lib.check(rcode, "gt_wn_stack_fwd")
        saved = (xs, ts, ss, acts_all, p, seed)
        if layers_only:
            return None, saved
        return conv_rows(acts_all, wn.pc_skipcat, rc, bias=wn.skip_bias, mask=True), saved
    if affine is not None:                            # the per-layer kernels read materialised rows
        cond, cond_per_row = cond_rows(aff_sig, (aff_w, aff_b)), True
    for i in range(n):
        ci = None if cond is None else cond[:, 2 * H * i:2 * H * (i + 1)]
        acts = acts_all[:, i * H:(i + 1) * H]
        last = i == n - 1
        if fused:
            il = wn.in_layers[i]
            rs = None if last else wn.res_skip_layers[i]
            t = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
            s = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
            xn = None if last else torch.empty(R, H, dtype=torch.bfloat16, device=dev)
            _ev = KERNEL_TIMER.start("wn_layer_fwd")
            rcode = L.gt_wn_layer_fwd(_lib.ptr(x), x.stride(0), _lib.ptr(il.pc.fwd), _lib.ptr(il.bias),
                                      _lib.ptr(ci), 0 if ci is None else ci.stride(0),
                                      _lib.ptr(rc.row0) if (ci is not None and not cond_per_row) else None,
                                      0 if (cond_per_row or ci is None) else rc.B, rc.Tp, _lib.ptr(rc.rowmask),
                                      _lib.ptr(acts), acts.stride(0), _lib.ptr(t), _lib.ptr(s), H,
                                      None if last else _lib.ptr(rs.pc_res.fwd),
                                      None if last else rs.bias.data_ptr(), _lib.ptr(xn), H,
                                      R, H, wn.kernel_size, float(p), int(seed + i), _lib.ptr(seed_word(dev)) if p > 0 else None,
                                      _lib.ptr(stamps.buf) if stamps else None, stamps.take("layer_last" if last else "layer") if stamps else 0,
                                      _lib.ptr(stamps.base) if stamps else None, _st(dev))
            KERNEL_TIMER.stop(_ev)
            _lib.check(rcode, "gt_wn_layer_fwd")
            ts.append(t); ss.append(s)
            if not last:
                x = xn
                xs.append(x)
            continue
        _, t, s = conv_rows(x, wn.in_layers[i].pc, rc, bias=wn.in_layers[i].bias, cond=ci, gate=True, out=acts,
                            drop_p=p, seed=seed + i, tag="in_layer_gate_conv", cond_per_row=cond_per_row)
        ts.append(t); ss.append(s)
        if i < n - 1:
            rs = wn.res_skip_layers[i]
            # rows [0,H) of the weight -> residual, rows [H,2H) -> skip (modules.py:166-168)
            x = conv_rows(acts, rs.pc_res, rc, bias=rs.bias[:H], addend=x, mask=True)
            xs.append(x)
    saved = (xs, ts, ss, acts_all, p, seed)
    if layers_only:                                   # the fused boundary kernel runs the skip GEMM (csrc/wn_boundary.hip)
        return None, saved
    out = conv_rows(acts_all, wn.pc_skipcat, rc, bias=wn.skip_bias, mask=True)
    return out, saved


def wn_bwd(rc, wn, saved, dskip, want_dcond=False, cond_per_row=False, dacts_skip=None, affine_grads=None):
    """dskip: [R,H] bf16, the MASKED gradient of the wn output (= d skip of every layer, since out = skip*mask).
    dacts_skip (fused layers only): dskip @ [W_skip_0 | ..] [R, n*H] when the caller already has it (boundary kernel).
    Returns (dh0 [R,H] bf16 masked, {param: grad}, dcond); dcond is [B, 2*H*n] (per-utterance sums) or, with
    cond_per_row, the per-frame gradient [R, 2*H*n] fp32.
    affine_grads = (sig [R, 2], dw [O], db [O]): the conditioning was modules.WNP's affine map (wn_fwd(affine=...)); the gradients of
    its two parameter vectors are ACCUMULATED into dw / db (gt_cond_affine_grads, straight from the gate backward's rows) and
    dcond comes back None — no [R, 2*H*n] fp32 gradient rows, no reductions on the host side."""
    if affine_grads is not None:
        want_dcond, cond_per_row = True, True
    if _fused_ok(wn):
        dh0, grads, dcond = _wn_bwd_fused(rc, wn, saved, dskip, want_dcond, cond_per_row, dacts_skip, affine_grads)
    else:
        assert dacts_skip is None
        dh0, grads, dcond = _wn_bwd_unfused(rc, wn, saved, dskip, want_dcond, cond_per_row)
    if affine_grads is not None and dcond is not None:       # a path that produced gradient rows: reduce them the old way
        g = cond_affine_grads(dcond, affine_grads[0])
        affine_grads[1].add_(g[0]); affine_grads[2].add_(g[1])
        dcond = None
    return dh0, grads, dcond


def _dcond_store(rc, dcond, i, H, src, cond_per_row):
    if cond_per_row:
        dcond[:, 2 * H * i:2 * H * (i + 1)] = src
    else:
        rc.utt_sum(src, dcond[:, 2 * H * i:2 * H * (i + 1)])


def _wn_bwd_fused(rc, wn, saved, dskip, want_dcond, cond_per_row, dacts_skip=None, affine_grads=None):
    """One kernel per layer boundary (gt_wn_layer_bwd): the k=5 data gradient of layer i+1's in_layer gives dX_{i+1} (the
    gradient at x_{i+1}); on that tile the residual 1x1's data gradient + the skip-path gradient + the gate backward of
    layer i follow, and d pre_i leaves for the next launch and for the weight gradients."""
    L = _lib.lib()
    xs, ts, ss, acts_all, p, seed = saved
    R, H = dskip.shape
    dev = dskip.device
    n = wn.n_layers
    grads = {}
    use_stack = getattr(wn, "stack_bwd", True) and n <= 4 and _stack_pays(R, n, dev)
    dcond = None if (not want_dcond or (use_stack and affine_grads is not None)) else \
        torch.empty(R if cond_per_row else rc.B, 2 * H * n, dtype=torch.float32, device=dev)
    need_c = want_dcond and p > 0                     # cond is added after the dropout: its gradient is d pre BEFORE the mask
    # skip path of every layer at once: dskip @ [W_skip_0 | ... | W_skip_{n-1}]  ->  [R, n*H]
    if dacts_skip is None:
        dacts_skip = conv_rows(dskip, wn.pc_skipcat, rc, dgrad=True)
    if use_stack:
        # the whole data-gradient chain in ONE launch (csrc/wn_stack.hip), then the weight-gradient jobs on what it wrote
        import ctypes
        bf = dict(dtype=torch.bfloat16, device=dev)
        dpre = [torch.empty(R, 2 * H, **bf) for _ in range(n)]
        dpre_c = [torch.empty(R, 2 * H, **bf) if need_c else None for _ in range(n)]
        dxs = [torch.empty(R, H, **bf) for _ in range(n)]                  # dxs[0] = gradient at the WaveNet's input
        pad = [None] * (4 - n)
        args = _lib.fill_args(
            _lib.WnStackBwdArgs, via_skip=dacts_skip, ldvs=dacts_skip.stride(0), gate_t=list(ts) + pad, gate_s=list(ss) + pad,
            w_in_d=[il.pc.dgrad for il in wn.in_layers] + pad, w_res_d=[rs.pc_res.dgrad for rs in wn.res_skip_layers[:n - 1]] + [None] + pad,
            rowmask=rc.rowmask, dpre=dpre + pad, dpre_c=dpre_c + pad, dx=dxs + pad, R=R, H=H, taps=wn.kernel_size, n_layers=n,
            drop_p=float(p), drop_seed=int(seed), seed_dev=seed_word(dev) if p > 0 else None)
        _ev = KERNEL_TIMER.start("wn_stack_bwd")
        rcode = L.gt_wn_stack_bwd(ctypes.byref(args), _st(dev))
        KERNEL_TIMER.stop(_ev)
        _lib.check(rcode, "gt_wn_stack_bwd")
        for i in reversed(range(n)):
            acts = acts_all[:, i * H:(i + 1) * H]
            if i == n - 1:
                grads.update(conv_param_grads(wn.res_skip_layers[i], acts, dskip, R))
            else:
                grads.update(conv_param_grads(wn.res_skip_layers[i], acts, None, R, parts=[(dxs[i + 1], 0, H), (dskip, H, H)]))
            grads.update(conv_param_grads(wn.in_layers[i], xs[i], dpre[i], R))
            if want_dcond and affine_grads is None:
                _dcond_store(rc, dcond, i, H, dpre_c[i] if need_c else dpre[i], cond_per_row)
        if affine_grads is not None:
            src = (dpre_c if need_c else dpre) + [None] * (4 - n)
            _lib.check(L.gt_cond_affine_grads(_lib.ptr(src[0]), _lib.ptr(src[1]), _lib.ptr(src[2]), _lib.ptr(src[3]), 2 * H,
                                              _lib.ptr(affine_grads[0]), _lib.ptr(affine_grads[1]), _lib.ptr(affine_grads[2]), R, H, n, _st(dev)),
                       "gt_cond_affine_grads")
        return dxs[0], grads, dcond
    # top layer: no residual output, d acts = skip path only
    i = n - 1
    dpre = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev)
    dpre_c = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev) if need_c else None
    via = dacts_skip[:, i * H:(i + 1) * H]
    _lib.check(L.gt_gate_bwd(_lib.ptr(via), via.stride(0), _lib.ptr(ts[i]), _lib.ptr(ss[i]), ts[i].stride(0), _lib.ptr(dpre), 2 * H,
                             _lib.ptr(dpre_c), R, H, float(p), int(seed + i), _lib.ptr(seed_word(dev)) if p > 0 else None, _st(dev)),
               "gt_gate_bwd")
    grads.update(conv_param_grads(wn.res_skip_layers[i], acts_all[:, i * H:(i + 1) * H], dskip, R))
    grads.update(conv_param_grads(wn.in_layers[i], xs[i], dpre, R))
    if want_dcond:
        _dcond_store(rc, dcond, i, H, dpre_c if need_c else dpre, cond_per_row)
    dX = None                                          # gradient at x_{i+1} (None above the top layer)
    for i in reversed(range(n - 1)):
        rs = wn.res_skip_layers[i]
        nxt = wn.in_layers[i + 1]
        dXn = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
        dpre_i = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev)
        dpre_ci = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev) if need_c else None
        via = dacts_skip[:, i * H:(i + 1) * H]
        _lib.check(L.gt_wn_layer_bwd(_lib.ptr(dpre), 2 * H, _lib.ptr(nxt.pc.dgrad), _lib.ptr(dX), H, _lib.ptr(rc.rowmask),
                                     _lib.ptr(dXn), H, _lib.ptr(rs.pc_res.dgrad), _lib.ptr(via), via.stride(0),
                                     _lib.ptr(ts[i]), _lib.ptr(ss[i]), H, _lib.ptr(dpre_i), _lib.ptr(dpre_ci), 2 * H, R, H, wn.kernel_size,
                                     float(p), int(seed + i), _lib.ptr(seed_word(dev)) if p > 0 else None, None, 0, None, _st(dev)),
                   "gt_wn_layer_bwd")
        acts = acts_all[:, i * H:(i + 1) * H]
        grads.update(conv_param_grads(rs, acts, None, R, parts=[(dXn, 0, H), (dskip, H, H)]))
        grads.update(conv_param_grads(wn.in_layers[i], xs[i], dpre_i, R))
        if want_dcond:
            _dcond_store(rc, dcond, i, H, dpre_ci if need_c else dpre_i, cond_per_row)
        dX, dpre = dXn, dpre_i
    # d x_0 = dgrad(in_layer_0) + (residual path), through the mask of x_0's producer: the same kernel without its second stage
    dh0 = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
    _lib.check(L.gt_wn_layer_bwd(_lib.ptr(dpre), 2 * H, _lib.ptr(wn.in_layers[0].pc.dgrad), _lib.ptr(dX), H, _lib.ptr(rc.rowmask),
                                 _lib.ptr(dh0), H, None, None, 0, None, None, 0, None, None, 0, R, H, wn.kernel_size, 0.0, 0, None, None, 0, None,
                                 _st(dev)), "gt_wn_layer_bwd")
    return dh0, grads, dcond


def _wn_bwd_unfused(rc, wn, saved, dskip, want_dcond=False, cond_per_row=False):
    """round 1's launch sequence (two GEMM kernels per layer), kept as the reference the fused path is tested against"""
    L = _lib.lib()
    xs, ts, ss, acts_all, p, seed = saved
    R, H = dskip.shape
    dev = dskip.device
    n = wn.n_layers
    grads = {}
    dcond = None if not want_dcond else torch.empty(R if cond_per_row else rc.B, 2 * H * n, dtype=torch.float32, device=dev)
    # skip path of every layer at once: dskip @ [W_skip_0 | ... | W_skip_{n-1}]  ->  [R, n*H]
    dacts_skip = conv_rows(dskip, wn.pc_skipcat, rc, dgrad=True)
    dres = None             # gradient arriving at x_{i+1} (masked), i.e. at res_i's output
    for i in reversed(range(n)):
        rs = wn.res_skip_layers[i]
        acts = acts_all[:, i * H:(i + 1) * H]
        via_skip = dacts_skip[:, i * H:(i + 1) * H]
        dpre_c = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev) if (want_dcond and p > 0) else None
        fused = i < n - 1 and dpre_c is None
        if fused:
            # d acts = d res @ W_res + (skip path), pushed straight through the gate in the GEMM's epilogue
            dpre = conv_rows(dres, rs.pc_res, rc, dgrad=True, addend=via_skip, gate=2, gate_t=ts[i], gate_s=ss[i],
                             drop_p=p, seed=seed + i)
        elif i < n - 1:
            dacts = conv_rows(dres, rs.pc_res, rc, dgrad=True, addend=via_skip)
        else:
            dacts = via_skip
        if i < n - 1:
            grads.update(conv_param_grads(rs, acts, None, R, parts=[(dres, 0, H), (dskip, H, H)]))
        else:
            grads.update(conv_param_grads(rs, acts, dskip, R))
        if not fused:
            dpre = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev)
            _lib.check(L.gt_gate_bwd(_lib.ptr(dacts), dacts.stride(0), _lib.ptr(ts[i]), _lib.ptr(ss[i]), ts[i].stride(0),
                                     _lib.ptr(dpre), 2 * H, _lib.ptr(dpre_c), R, H, float(p), int(seed + i),
                                     _lib.ptr(seed_word(dev)) if p > 0 else None, _st(dev)), "gt_gate_bwd")
        grads.update(conv_param_grads(wn.in_layers[i], xs[i], dpre, R))
        if want_dcond:
            _dcond_store(rc, dcond, i, H, dpre_c if dpre_c is not None else dpre, cond_per_row)
        # d x_i = dgrad(in_layer) + (residual path), then through the mask of x_i's producer
        dres = conv_rows(dpre, wn.in_layers[i].pc, rc, dgrad=True, addend=dres, mask=True)
    return dres, grads, dcond


# ----------------------------------------------------------------------------- coupling block
def contour_rows(rc, c, B, T):
    """[b,1,t] pitch / energy contour -> [R, 2] fp32 rows (column = frame parity): the time squeeze of modules.py:353-362
    for one channel, on the squeezed rows context rc (T = un-squeezed frames covered, even)."""
    if c is None:
        return None
    L = _lib.lib()
    cc = c.detach().float().reshape(B, 1, -1)[:, :, :T].contiguous()
    assert cc.shape[2] == T, "pitch / energy must cover the mel frames"
    rows = torch.empty(rc.R, 2, dtype=torch.float32, device=cc.device)
    _lib.check(L.gt_squeeze_rows_f32(_lib.ptr(cc), _lib.ptr(rows), _lib.ptr(rc.lengths), B, 1, T, rc.Tp, _lib.ptr(rc.row0),
                                     _st(cc.device)), "gt_squeeze_rows_f32")
    return rows


def cond_rows(sig, aff_b):
    """per-frame conditioning rows [R, 2*O] = w * contour + b, laid out like the squeezed cond_layer1 output
    (channel = parity * O + c); layer i of the WNP reads columns [2*H*i, 2*H*(i+1)).  aff_b = (w [O], b [O])."""
    if sig is None:
        return None
    w, b = aff_b[0].detach().float(), aff_b[1].detach().float()
    return torch.addcmul(b[None, None, :], sig[:, :, None], w[None, None, :]).reshape(sig.shape[0], -1)


def cond_affine_grads(dc, sig):
    """gradient of cond_rows w.r.t. (w, b): d w = sum dcond * contour, d b = sum dcond over (frame, parity) -> [2, O]"""
    O = dc.shape[1] // 2
    d3 = dc.view(dc.shape[0], 2, O)
    return torch.stack([(d3 * sig[:, :, None]).sum((0, 1)), d3.sum((0, 1))])


def prosody_chain(cb, econd, pcond):
    """The per-frame conditioned WaveNets that follow cb.wn, in the reference's order (attentions.py:153-154):
    wn_energy then wn_pitch; each is the identity when its conditioning is None (modules.py:323-324)."""
    return [(w, c) for w, c in ((getattr(cb, "wn_energy", None), econd), (getattr(cb, "wn_pitch", None), pcond)) if c is not None]


def coupling_fwd(rc, cb, x, x0_bf16, cond, logdet, train, seed, econd=None, pcond=None):
    """attentions.CouplingBlock.forward on rows.  x [R,C] fp32, x0_bf16 = bf16(x[:, :C/2]).
    econd / pcond: [R, 2*H*n] fp32 per-frame conditioning of wn_energy / wn_pitch (cond_layer1 output, squeezed)."""
    L = _lib.lib()
    dev = x.device
    R, C = x.shape
    h0 = conv_rows(x0_bf16, cb.start.pc, rc, bias=cb.start.bias, mask=True)
    wn_out, wn_saved = wn_fwd(rc, cb.wn, h0, cond, train, seed)
    pros_saved = []
    for k, (wnp, c) in enumerate(prosody_chain(cb, econd, pcond)):
        wn_out, sv = wn_fwd(rc, wnp, wn_out, c, train, seed + 4 * (k + 1), cond_per_row=True)
        pros_saved.append(sv)
    if pros_saved:
        wn_saved = (wn_saved, pros_saved)
    out = conv_rows(wn_out, cb.end.pc, rc, bias=cb.end.bias, out_f32=True)      # [R,C] = [m | logs]
    z = torch.empty_like(x)
    _lib.check(L.gt_coupling_fwd(_lib.ptr(out), _lib.ptr(x), _lib.ptr(z), _lib.ptr(rc.rowmask), _lib.ptr(logdet),
                                 rc.B, R, C, rc.Tp, _lib.ptr(rc.row0), int(cb.sigmoid_scale), _st(dev)), "gt_coupling_fwd")
    return z, (x, x0_bf16, h0, wn_out, wn_saved, out)


def coupling_bwd(rc, cb, saved, dz, dlogdet, want_dcond=False, econd=False, pcond=False):
    """econd / pcond: whether wn_energy / wn_pitch ran in the forward; returns (dx, grads, dcond, [d econd, d pcond])."""
    L = _lib.lib()
    x, x0_bf16, h0, wn_out, wn_saved, out = saved
    chain = prosody_chain(cb, econd or None, pcond or None)
    pros_saved = []
    if chain:
        wn_saved, pros_saved = wn_saved
    dev = x.device
    R, C = x.shape
    dx = torch.empty_like(x)
    dout = torch.empty(R, C, dtype=torch.bfloat16, device=dev)
    _lib.check(L.gt_coupling_bwd(_lib.ptr(out), _lib.ptr(x), _lib.ptr(dz), _lib.ptr(dlogdet), _lib.ptr(rc.rowmask),
                                 _lib.ptr(dx), _lib.ptr(dout), rc.B, R, C, rc.Tp, _lib.ptr(rc.row0), int(cb.sigmoid_scale), _st(dev)),
               "gt_coupling_bwd")
    grads = conv_param_grads(cb.end, wn_out, dout, R)
    dskip = conv_rows(dout, cb.end.pc, rc, dgrad=True, mask=True)                # d(wn out) * mask = d skip
    dpros = {}
    for (wnp, _), sv in zip(reversed(chain), reversed(pros_saved)):              # each WN's masked input gradient is the
        dskip, gp, dc = wn_bwd(rc, wnp, sv, dskip, True, cond_per_row=True)      # previous WN's d skip
        grads.update(gp)
        dpros[id(wnp)] = dc
    dh0, g2, dcond = wn_bwd(rc, cb.wn, wn_saved, dskip, want_dcond)
    grads.update(g2)
    grads.update(conv_param_grads(cb.start, x0_bf16, dh0, R))                    # dh0 is already masked
    # d x0 = (identity path, already in dx[:, :C/2]) + dgrad(start): added in the GEMM's epilogue, in place
    conv_rows(dh0, cb.start.pc, rc, dgrad=True, addend=dx[:, :C // 2], out=dx[:, :C // 2])
    if econd or pcond:
        return dx, grads, dcond, [dpros.get(id(getattr(cb, "wn_energy", None))) if econd else None,
                                  dpros.get(id(getattr(cb, "wn_pitch", None))) if pcond else None]
    return dx, grads, dcond


# ----------------------------------------------------------------------------- fused between-WaveNets kernels
# (whether a WaveNet runs as ONE launch per direction or one per layer is a property of the module: modules.WN.set_stack)
BOUNDARY_TRACE = None        # dev (tools/wn_boundary_bench.py): a list collects the (entry name, args struct, keep-alive) of every launch


class _BlockState:
    """what one flow block keeps for the backward on the fused path"""
    __slots__ = ("x_in", "y", "x0", "h0", "wn_saved", "wn_out", "logs_raw", "z", "scal", "w_ic", "chain")


def _ptr_table(dec):
    """device arrays of the blocks' ActNorm.logs / InvConvNear.weight pointers (gt_flow_scalars_multi); rebuilt when a
    parameter's storage moves"""
    ans = [dec.flows[3 * b] for b in range(dec.n_blocks)]
    ics = [dec.flows[3 * b + 1] for b in range(dec.n_blocks)]
    key = tuple(a.logs.data_ptr() for a in ans) + tuple(i.weight.data_ptr() for i in ics)
    tab = getattr(dec, "_scal_table", None)
    if tab is None or tab[0] != key:
        assert not torch.cuda.is_current_stream_capturing(), "flow-scalar pointer table is built outside graph capture"
        dev = ans[0].logs.device
        tab = (key, torch.tensor([a.logs.data_ptr() for a in ans], dtype=torch.int64).to(dev),
               torch.tensor([i.weight.data_ptr() for i in ics], dtype=torch.int64).to(dev))
        object.__setattr__(dec, "_scal_table", tab)
    return tab[1], tab[2]


def _grad_ptr_table(dec, tensors):
    """device array of the gradient accumulators' addresses (gt_boundary_param_reduce).  Under train.Trainer these are slices of the
    flat gradient buffer: the table is built once, outside graph capture (the eager warm-up steps), and reused while they stay put."""
    key = tuple(t.data_ptr() for t in tensors)
    tab = getattr(dec, "_pg_table", None)
    if tab is None or tab[0] != key:
        assert not torch.cuda.is_current_stream_capturing(), "gradient pointer table is built outside graph capture"
        tab = (key, torch.tensor(key, dtype=torch.int64).to(tensors[0].device))
        object.__setattr__(dec, "_pg_table", tab)
    return tab[1]


def flow_scalars_all(dec):
    """scal [n_blocks, 18] = {sum logs, logdet W, W^-T} of every block in ONE launch, and the blocks' (contiguous) 4x4
    weights.  A weight that is a strided view (a hand-filled test module; trained parameters are slices of the flat
    buffer) is copied and takes the one-pair-per-launch route."""
    nb = dec.n_blocks
    Ws = [dec.flows[3 * b + 1].weight.detach() for b in range(nb)]
    lgs = [dec.flows[3 * b].logs.detach() for b in range(nb)]
    if all(w.is_contiguous() for w in Ws) and all(l.is_contiguous() for l in lgs):
        lp, wp = _ptr_table(dec)
        C = dec.flows[0].channels
        scal = torch.empty(nb, 18, dtype=torch.float32, device=lp.device)
        _lib.check(_lib.lib().gt_flow_scalars_multi(_lib.ptr(lp), _lib.ptr(wp), C, _lib.ptr(scal), nb, _st(lp.device)),
                   "gt_flow_scalars_multi")
        return scal, Ws
    Ws = [w.contiguous() for w in Ws]
    return torch.stack([flow_scalars(l, w) for l, w in zip(lgs, Ws)]), Ws


def _prefetch_list(wn, attr, extra=()):
    """(pointers, byte counts) of a WaveNet's weight images for the boundary launches' prefetch workgroups (attr: "fwd" — the forward
    images — or "dgrad"): the four in_layers and the residual 1x1s, what the whole-WaveNet launch that follows streams.  extra:
    activations that launch reads and that were written long ago (the backward's saved tanh / sigmoid halves)."""
    n = wn.n_layers
    imgs = [getattr(il.pc, attr) for il in wn.in_layers] + [getattr(rs.pc_res, attr) for rs in wn.res_skip_layers[:n - 1]] + list(extra)
    imgs = [t for t in imgs if t is not None][:16]
    return imgs + [None] * (16 - len(imgs)), [t.numel() * t.element_size() // 16 * 16 for t in imgs] + [0] * (16 - len(imgs))


def block_chain(cb, cond, esig=None, eaff_b=None, psig=None, paff_b=None):
    """The WaveNets of one coupling block in the reference's order (attentions.py:152-154: wn, wn_energy, wn_pitch; the latter two
    are the identity when their contour is None, modules.py:323-324) with what conditions each: [(module, cond, affine)]."""
    chain = [(cb.wn, cond, None)]
    for w, sig, aff in ((getattr(cb, "wn_energy", None), esig, eaff_b), (getattr(cb, "wn_pitch", None), psig, paff_b)):
        if w is not None and sig is not None:
            chain.append((w, None, (sig, aff[0], aff[1])))
    return chain


def decoder_fwd_fused(rc, dec, rows, conds, logdet, train, seed, y_bct=None, z_bct=None, esig=None, eaff=None, psig=None, paff=None):
    """The decoder's flow chain with ONE kernel between consecutive WaveNets (gt_wn_boundary_fwd: tail of block b-1 + head
    of block b) and one kernel per WaveNet layer: n_blocks * (n_layers + 1) + 1 launches.  rows [R, C] fp32 (squeezed mel),
    conds[b]: [B, 2*H*n] or None.  Returns (z rows, per-block saved state).
    y_bct / z_bct ([B, C/2, T] fp32, T even, z_bct pre-zeroed): the decoder's input / output at the public boundary — the first
    launch then squeezes (rows = None) and the last one unsqueezes (the returned z rows are None)."""
    L = _lib.lib()
    dev = rc.rowmask.device
    R, C = rc.R, 2 * dec.in_channels
    H, nb, n = dec.hidden_channels, dec.n_blocks, dec.n_layers
    scal, Ws = flow_scalars_all(dec)
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    blocks = []
    for b in range(nb + 1):
        kw = dict(rowmask=rc.rowmask, R=R, H=H, C=C, n_layers=n, logdet=logdet)
        if b > 0:                                          # tail of block b-1: on the LAST WaveNet of its chain
            sv, cbp = blocks[b - 1], dec.flows[3 * (b - 1) + 2]
            last_wn = sv.chain[-1][0]
            acts_all = sv.wn_saved[3]
            sv.wn_out = torch.empty(R, H, **bf)
            sv.logs_raw = torch.empty(R, C // 2, **f32)
            last_out = b == nb and z_bct is not None
            sv.z = None if last_out else torch.empty(R, C, **f32)
            kw.update(acts=acts_all, ldacts=acts_all.stride(0), w_skip=last_wn.pc_skipcat_frag.fwd, b_skip=last_wn.skip_bias,
                      w_end=cbp.end.pc_frag.fwd, b_end=cbp.end.bias, ks_end=cbp.end.pc_frag.Kp_f // 16, y=sv.y, wn_out=sv.wn_out,
                      logs_raw=sv.logs_raw, z=sv.z, rowutt=rc.rowutt, sigmoid_scale=int(cbp.sigmoid_scale))
            if last_out:
                kw.update(z_bct=z_bct, T=z_bct.shape[2], rowbatch=rc.rowbatch, rowframe=rc.rowframe, len=rc.lengths)
        if b < nb:                                         # head of block b
            an, ic, cb = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2]
            st = _BlockState()
            if b == 0 and rows is None:
                rows = torch.empty(R, C, **f32)            # the first launch writes the squeezed rows (ActNorm's input, for the backward)
                kw.update(y_bct=y_bct, T=y_bct.shape[2], rowbatch=rc.rowbatch, rowframe=rc.rowframe, z=rows)
            elif b == 0:
                kw.update(x_in=rows)
            st.x_in = rows if b == 0 else blocks[b - 1].z
            st.y = torch.empty(R, C, **f32)
            st.x0 = torch.empty(R, C // 2, **bf)
            st.h0 = torch.empty(R, H, **bf)
            st.scal, st.w_ic = scal[b], Ws[b]
            kw.update(an_logs=an.logs, an_bias=an.bias, w_ic=st.w_ic, scal=st.scal, len=rc.lengths, B=rc.B, y_next=st.y, y0_bf16=st.x0,
                      w_start=cb.start.pc_frag.fwd, b_start=cb.start.bias, ks_start=cb.start.pc_frag.Kp_f // 16, h_next=st.h0)
            if _fused_ok(cb.wn):                       # the WaveNet launch that follows: its images are read once by this launch's spare CUs
                kw["pf_ptr"], kw["pf_bytes"] = _prefetch_list(cb.wn, "fwd")
            blocks.append(st)
        args = _lib.fill_args(_lib.BoundaryFwdArgs, **kw)
        import ctypes
        if BOUNDARY_TRACE is not None:
            BOUNDARY_TRACE.append(("gt_wn_boundary_fwd", args, kw))
        _ev = KERNEL_TIMER.start("wn_boundary_fwd")
        rcode = L.gt_wn_boundary_fwd(ctypes.byref(args), _st(dev))
        KERNEL_TIMER.stop(_ev)
        _lib.check(rcode, "gt_wn_boundary_fwd")
        if b < nb:
            # the block's WaveNets: wn (speaker vector) [-> wn_energy -> wn_pitch (affine per-frame conditioning)]; between two of them
            # only the skip GEMM + mask (one launch); the last one's gated activations go straight into the next boundary launch
            st = blocks[b]
            chain = block_chain(dec.flows[3 * b + 2], conds[b], esig, None if eaff is None else eaff[b], psig, None if paff is None else paff[b])
            h, st.chain = st.h0, []
            for k, (w, c, aff) in enumerate(chain):
                last = k == len(chain) - 1
                out, sv_k = wn_fwd(rc, w, h, c, train, seed + 16 * b + 4 * k, layers_only=last, affine=aff)
                st.chain.append((w, sv_k, aff))
                h = out
            st.wn_saved = st.chain[-1][1]
    return blocks[-1].z, blocks


def decoder_bwd_fused(rc, dec, blocks, drows, dlogdet, has_cond, dz_bct=None, dx_bct=None, deaff=None, dpaff=None):
    """Backward of decoder_fwd_fused: gt_wn_boundary_bwd between the WaveNets' layer kernels.  drows [R, C] fp32 = gradient
    of the decoder's output rows; returns (d input rows [R, C], {param: grad}, [dcond per block]).
    dz_bct / dx_bct ([B, C/2, T] fp32, T even, dx_bct pre-zeroed): the gradients at the public boundary — the first launch then
    squeezes dz (drows = None), the last one unsqueezes the input gradient (returned rows are None)."""
    import ctypes
    L = _lib.lib()
    dev = rc.rowmask.device
    R, C = rc.R, 2 * dec.in_channels
    H, nb, n = dec.hidden_channels, dec.n_blocks, dec.n_layers
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    grads, dconds = {}, [None] * nb
    # ActNorm / InvConvNear parameter gradients: one row of partial sums per workgroup and launch, added up by ONE launch after the
    # pass (gt_boundary_param_reduce) — as atomics, 152 workgroups on the same 336 addresses cost ~10 us of every 33 us launch
    n_wg, PG = (R + 63) // 64, L.gt_boundary_param_partials()
    pg = torch.empty(nb, n_wg * PG, **f32)
    pg_dst = []
    dx_prev = None                                         # [d z0 | d y1] of the block whose WaveNet backward runs next
    dh0 = None
    tail = None                                            # (dout, dwn_out, via_skip) of that block
    for b in range(nb, -1, -1):
        kw = dict(rowmask=rc.rowmask, R=R, H=H, C=C, n_layers=n)
        dx_out = None if (b == 0 and dx_bct is not None) else torch.empty(R, C, **f32)
        if b < nb:                                         # head of block b: its WaveNet backward has just produced dh0
            an, ic, cb, sv = dec.flows[3 * b], dec.flows[3 * b + 1], dec.flows[3 * b + 2], blocks[b]
            dlogs = grad_accumulator(an.logs, (C,))
            dbias = grad_accumulator(an.bias, (C,))
            dW = grad_accumulator(ic.weight, (16,))
            kw.update(dh=dh0, w_start_d=cb.start.pc_frag.dgrad, ks_start_d=cb.start.pc_frag.Kp_d // 16, dx_in=dx_prev, x=sv.x_in,
                      an_logs=an.logs, an_bias=an.bias, w_ic=sv.w_ic, scal=sv.scal, len=rc.lengths, B=rc.B,
                      d_an_logs=dlogs, d_an_bias=dbias, d_w_ic=dW, dlogdet=dlogdet, pg_partial=pg[b])
            pg_dst.append((b, dlogs, dbias, dW))
            grads.update({an.logs: dlogs.view_as(an.logs), an.bias: dbias.view_as(an.bias), ic.weight: dW.view_as(ic.weight)})
        elif dz_bct is not None:
            kw.update(dz_bct=dz_bct, T=dz_bct.shape[2], rowbatch=rc.rowbatch, rowframe=rc.rowframe, len=rc.lengths)
        else:
            kw.update(dz_in=drows)
        new_tail = None
        if b > 0:                                          # tail of block b-1
            cbp, svp = dec.flows[3 * (b - 1) + 2], blocks[b - 1]
            dout = torch.empty(R, C, **bf)
            dwn = torch.empty(R, H, **bf)
            via = torch.empty(R, n * H, **bf)
            kw.update(logs_raw=svp.logs_raw, y=svp.y, dlogdet=dlogdet, rowutt=rc.rowutt, sigmoid_scale=int(cbp.sigmoid_scale),
                      dout=dout, w_end_d=cbp.end.pc_frag.dgrad, ks_end_d=cbp.end.pc_frag.Kp_d // 16, dwn_out=dwn,
                      w_skip_d=svp.chain[-1][0].pc_skipcat_frag.dgrad, ks_skip_d=svp.chain[-1][0].pc_skipcat_frag.Kp_d // 16, via_skip=via,
                      ldvs=via.stride(0))
            if _fused_ok(svp.chain[-1][0]):
                # (the saved tanh / sigmoid halves the same launch reads — 30 MB written in the forward pass — were tried as `extra`: no
                # change in the step; the backward launch's cold cost is not where its 64 spare workgroups can reach in 28 us)
                kw["pf_ptr"], kw["pf_bytes"] = _prefetch_list(svp.chain[-1][0], "dgrad")
            new_tail = (dout, dwn, via)
        if dx_out is None:
            kw.update(dx_bct=dx_bct, T=dx_bct.shape[2], rowbatch=rc.rowbatch, rowframe=rc.rowframe)
        else:
            kw.update(dx_out=dx_out)
        args = _lib.fill_args(_lib.BoundaryBwdArgs, **kw)
        if BOUNDARY_TRACE is not None:
            BOUNDARY_TRACE.append(("gt_wn_boundary_bwd", args, kw))
        _ev = KERNEL_TIMER.start("wn_boundary_bwd")
        rcode = L.gt_wn_boundary_bwd(ctypes.byref(args), _st(dev))
        KERNEL_TIMER.stop(_ev)
        _lib.check(rcode, "gt_wn_boundary_bwd")
        if b == 0:
            pg_dst.sort(key=lambda t: t[0])
            tab = _grad_ptr_table(dec, [t for _, *ts in pg_dst for t in ts])
            _lib.check(L.gt_boundary_param_reduce(_lib.ptr(pg), n_wg, nb, _lib.ptr(tab), _st(dev)), "gt_boundary_param_reduce")
            return dx_out, grads, dconds
        # block b-1: end conv's parameter gradients, the WaveNet's backward, then the start conv's
        cbp, svp = dec.flows[3 * (b - 1) + 2], blocks[b - 1]
        dout, dwn, via = new_tail
        grads.update(conv_param_grads(cbp.end, svp.wn_out, dout, R))
        # the chain backwards: the masked input gradient of a WaveNet is the d skip of the one before it; the last one's skip-path
        # gradient came out of the boundary launch (via), the others compute theirs (one launch)
        dskip = dwn
        for k in reversed(range(len(svp.chain))):
            w, sv_k, aff = svp.chain[k]
            ag = None
            if aff is not None:
                dst = deaff if w is getattr(cbp, "wn_energy", None) else dpaff
                ag = (aff[0], dst[b - 1, 0], dst[b - 1, 1])
            dskip, g2, dc = wn_bwd(rc, w, sv_k, dskip, has_cond if k == 0 else False, dacts_skip=via if k == len(svp.chain) - 1 else None,
                                   affine_grads=ag)
            grads.update(g2)
            if k == 0:
                dconds[b - 1] = dc
        dh0 = dskip
        grads.update(conv_param_grads(cbp.start, svp.x0, dh0, R))
        dx_prev = dx_out


# ----------------------------------------------------------------------------- reverse flow (inference)
def actnorm_invconv_rev(rc, y, logs, bias, W, want_x0=True, scal=None):
    """InvConvNear^-1 then ActNorm^-1 on rows (modules.py:647-652, 592-594).  y: [R,C] fp32 -> x, bf16(x[:, :C/2]).
    scal: the pair's cached flow scalars (FlowSpecDecoder.store_inverse); computed here when None."""
    L = _lib.lib()
    dev = y.device
    R, C = y.shape
    x = torch.empty_like(y)
    x0 = torch.empty(R, C // 2, dtype=torch.bfloat16, device=dev) if want_x0 else None
    lg = logs.detach().reshape(-1).contiguous()
    bs = bias.detach().reshape(-1).contiguous()
    if scal is None:
        scal = flow_scalars(logs, W)
    _lib.check(L.gt_actnorm_invconv_rev(_lib.ptr(y), _lib.ptr(x), _lib.ptr(x0), C // 2, _lib.ptr(lg), _lib.ptr(bs), _lib.ptr(scal),
                                        _lib.ptr(rc.rowmask), R, C, _st(dev)), "gt_actnorm_invconv_rev")
    return x, x0


def coupling_rev(rc, cb, z, z0_bf16, cond, econd=None, pcond=None):
    """attentions.CouplingBlock.forward with reverse=True on rows: the same start / WN / end GEMMs as the forward
    (evaluation mode), then x1 = (z1 - m) * exp(-logs) * mask."""
    L = _lib.lib()
    dev = z.device
    R, C = z.shape
    h0 = conv_rows(z0_bf16, cb.start.pc, rc, bias=cb.start.bias, mask=True)
    wn_out, _ = wn_fwd(rc, cb.wn, h0, cond, False, 0)
    for wnp, c in prosody_chain(cb, econd, pcond):
        wn_out, _ = wn_fwd(rc, wnp, wn_out, c, False, 0, cond_per_row=True)
    out = conv_rows(wn_out, cb.end.pc, rc, bias=cb.end.bias, out_f32=True)
    x = torch.empty_like(z)
    _lib.check(L.gt_coupling_rev(_lib.ptr(out), _lib.ptr(z), _lib.ptr(x), _lib.ptr(rc.rowmask), R, C, int(cb.sigmoid_scale),
                                 _st(dev)), "gt_coupling_rev")
    return x
