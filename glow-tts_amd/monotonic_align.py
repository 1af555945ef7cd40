"""Drop-in for the reference's ``monotonic_align`` package, on the MI355X.

Mirrors reference monotonic_align/__init__.py:6-21 (``maximum_path(value, mask)``), whose
Cython core (core.pyx:9-45) is replaced by the HIP kernel ``gt_mas_f32`` (csrc/mas.hip)
called through the C-ABI.  No device->host round trip, no copies of the lattice.
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.GT_DT_F32, torch.int32: _lib.GT_DT_I32, torch.float16: _lib.GT_DT_F16,
       torch.bfloat16: _lib.GT_DT_BF16, torch.uint8: _lib.GT_DT_U8}


class MASResult:
    """Everything the kernel produces in one launch."""
    __slots__ = ("path", "durations", "frame2token", "status", "workspace")

    def __init__(self, path, durations, frame2token, status, workspace=None):
        self.path, self.durations, self.frame2token, self.status = path, durations, frame2token, status
        self.workspace = workspace          # int32 [B, T_x+1] row start columns (gt_mas_f32 workspace)


def maximum_path_lengths(value, t_x, t_y, mask=None, out_dtype=None, want_durations=False,
                         want_frame2token=False, validate=False, keep_workspace=False):
    """MAS from explicit lengths (the native form: no mask traffic).

    value: [b, t_x_max, t_y_max] float tensor on the GPU (fp32 is used as is; other float
           dtypes are widened to fp32 first, like ``astype(np.float32)`` in __init__.py:14).
    t_x, t_y: [b] int32 device tensors.
    mask:  optional [b, t_x_max, t_y_max]; when given the kernel evaluates value*mask
           (__init__.py:11) on the fly.
    Returns MASResult; ``path`` has dtype ``out_dtype`` (default value.dtype).
    """
    _lib.require_cuda(value, t_x, t_y, mask)
    L = _lib.lib()
    if value.dim() != 3:
        raise ValueError("value must be [b, t_x, t_y]")
    out_dtype = out_dtype or value.dtype
    if out_dtype not in _DT:
        raise TypeError(f"unsupported path dtype {out_dtype}")
    v = value.detach()
    if v.dtype != torch.float32:
        v = v.float()
    if v.stride(2) != 1:
        v = v.contiguous()
    B, T_x, T_y = v.shape
    m = None
    if mask is not None:
        m = mask.detach().to(torch.float32).expand_as(v)
        if m.stride() != v.stride():
            m = m.contiguous()
            v = v.contiguous()
    t_x = t_x.to(device=v.device, dtype=torch.int32).contiguous()
    t_y = t_y.to(device=v.device, dtype=torch.int32).contiguous()
    path = torch.empty((B, T_x, T_y), dtype=out_dtype, device=v.device)
    dur = torch.empty((B, T_x), dtype=torch.float32, device=v.device) if want_durations else None
    f2t = torch.empty((B, T_y), dtype=torch.int32, device=v.device) if want_frame2token else None
    status = torch.zeros((1,), dtype=torch.int32, device=v.device) if validate else None
    ws = None
    if B and T_x and T_y:
        ws_bytes = L.gt_mas_workspace_bytes(B, T_x, T_y)
        ws = torch.empty((ws_bytes // 4,), dtype=torch.int32, device=v.device)
        rc = L.gt_mas_f32(_lib.ptr(v), _lib.ptr(m), _lib.ptr(t_x), _lib.ptr(t_y), _lib.ptr(path),
                          _DT[out_dtype], _lib.ptr(dur), _lib.ptr(f2t), B, T_x, T_y,
                          v.stride(0), v.stride(1), _lib.ptr(ws), ws_bytes, _lib.ptr(status),
                          _lib.current_stream(v.device))
        if rc == -2:
            raise RuntimeError(f"gt_mas_f32: lattice [{T_x},{T_y}] exceeds the kernel's limits "
                               f"(T_x<=512, LDS {L.gt_mas_lds_bytes(T_x, T_y)} B > 160 KiB)")
        _lib.check(rc, "gt_mas_f32")
    if validate:
        st = int(status.item())
        if st & 1:
            raise ValueError("maximum_path: t_x > t_y for some utterance (the reference reads out of "
                             "bounds here, core.pyx:34); refusing")
        if st & 2:
            raise ValueError("maximum_path: a length is negative or exceeds the lattice")
    starts = ws[:B * (T_x + 1)].view(B, T_x + 1) if (keep_workspace and ws is not None) else None
    return MASResult(path, dur, f2t, status, starts)


def result_from_path(path, t_x, t_y):
    """A MASResult for a GIVEN monotonic path [b, t_x_max, t_y_max] (entries {0, 1}, one 1 per valid frame): durations, the
    frame -> token map (-1 on padded frames) and the per-row start columns, as gt_mas_f32 lays them out.  Test hook
    (FlowGenerator.forward(path=...)): the reference's own alignment instead of the one searched on this model's lattice."""
    B, T_x, T_y = path.shape
    p = path.detach().to(torch.float32)
    dur = p.sum(2)
    yy = torch.arange(T_y, device=p.device)[None, :]
    f2t = p.argmax(1).to(torch.int32)
    f2t = torch.where(yy < t_y.to(p.device)[:, None], f2t, torch.full_like(f2t, -1))
    starts = torch.zeros(B, T_x + 1, dtype=torch.int32, device=p.device)
    starts[:, 1:] = torch.cumsum(dur, 1).to(torch.int32)
    xx = torch.arange(T_x + 1, device=p.device)[None, :]
    starts = torch.where(xx >= t_x.to(p.device)[:, None], t_y.to(p.device, torch.int32)[:, None].expand_as(starts), starts)
    return MASResult(path, dur, f2t.contiguous(), None, starts.contiguous())


def lengths_from_mask(mask):
    """t_x, t_y as reference monotonic_align/__init__.py:18-19 derives them, on the device."""
    _lib.require_cuda(mask)
    L = _lib.lib()
    m = mask.detach().to(torch.float32)
    if m.stride(2) != 1:
        m = m.contiguous()
    B, T_x, T_y = m.shape
    t_x = torch.zeros((B,), dtype=torch.int32, device=m.device)
    t_y = torch.zeros((B,), dtype=torch.int32, device=m.device)
    if B and T_x and T_y:
        _lib.check(L.gt_mas_lengths_from_mask_f32(_lib.ptr(m), _lib.ptr(t_x), _lib.ptr(t_y), B, T_x, T_y,
                                                  m.stride(0), m.stride(1), _lib.current_stream(m.device)),
                   "gt_mas_lengths_from_mask_f32")
    return t_x, t_y


def maximum_path(value, mask, validate=False):
    """Same contract as reference monotonic_align.maximum_path (__init__.py:6-21):

    value: [b, t_x, t_y], mask: [b, t_x, t_y]  ->  path [b, t_x, t_y] on value's device, in
    value's dtype, entries {0,1}.  Inputs are not modified.  Lengths come from the mask's first
    column / row and the DP runs on value*mask, exactly like the reference.
    """
    t_x, t_y = lengths_from_mask(mask)
    return maximum_path_lengths(value, t_x, t_y, mask=mask, validate=validate).path
