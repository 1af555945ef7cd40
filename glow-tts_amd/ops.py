"""Thin host-side wrappers over the C-ABI kernels (rows layout helpers, packed conv weights).

PyTorch is used here only for device memory, streams and trivial index plumbing (building the
row mask); all arithmetic of the hot path happens in libglowtts_hip.so.
"""
import torch

from . import _lib

HALO = 2  # GT_HALO in include/glowtts_hip.h


def _round_up(x, m):
    return (x + m - 1) // m * m


class _KernelTimer:
    """HIP events around every launch carrying a given tag (bench.py: live duration of the dominant
    kernel inside the timed region, on the stream the kernel is launched on)."""

    def __init__(self):
        self.tag, self.pairs = None, []

    def enable(self, tag):
        self.tag, self.pairs = tag, []

    def disable(self):
        self.tag = None

    def start(self, tag):
        if self.tag is None or tag != self.tag:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def stop(self, e0):
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.append((e0, e1))

    def collect(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        out = {"ms": ms, "count": len(self.pairs)}
        self.tag, self.pairs = None, []
        return out


KERNEL_TIMER = _KernelTimer()


class KernelStamps:
    """Device-side begin / end stamps of the fused WaveNet-layer forward launches (wall_clock64 ticks): the kernel's duration
    INSIDE a replayed HIP graph, where there is no room for event pairs between kernel nodes.  A RowsCtx carries it
    (`rc.stamps`); the step bumps `base` by `per_step` (one tiny launch), so every replay fills its own slots."""

    def __init__(self, device, per_step, max_steps=512):
        self.per_step, self.max_steps = per_step, max_steps
        n = per_step * max_steps
        init = torch.empty(2 * n, dtype=torch.int64)
        init[0::2] = 0                                   # start: workgroup 0's clock (plain store)
        init[1::2] = 0                                   # end: atomicMax over the workgroups
        self.buf = init.to(device)
        self.base = torch.zeros(1, dtype=torch.int32, device=device)
        self._next = 0
        self.kinds = {}                                  # slot within a step -> {kernel kinds that ever stamped it}
        self.steps_done = 0                              # host-side count of end_step() calls (captured steps count once per capture)

    def begin_step(self):
        """inside the step (captured with it): the slots of this execution start at the counter's current value"""
        self._next = 0

    def end_step(self):
        # the slot counter wraps inside the buffer: the kernels index stamps[2 * (slot + base)] unchecked (a run longer than
        # max_steps then overwrites its oldest slots instead of writing past the allocation)
        self.base.add_(self.per_step).remainder_(self.per_step * self.max_steps)
        self.steps_done += 1

    def take(self, kind=""):
        """the next slot of the step; `kind` names the kernel that will stamp it (bench.py reads `kinds` to tell the WaveNet stack
        launches from the per-layer ones a long batch falls back to)"""
        k = self._next
        self._next += 1
        assert k < self.per_step
        self.kinds.setdefault(k, set()).add(kind)
        return k

    def durations_ticks(self):
        """[steps executed, per_step] int64 tensor of (max end - min start), on the host"""
        n = int(self.base.item()) // self.per_step
        b = self.buf[: 2 * n * self.per_step].cpu().view(n, self.per_step, 2)
        return (b[:, :, 1] - b[:, :, 0])

_SEED = {}


def seed_word(device):
    """Device-resident uint32 mixed into every dropout seed (seed_dev of the C-ABI).  Bumping it (even
    from inside a captured HIP graph) gives fresh dropout masks without changing any kernel argument."""
    key = str(device)
    t = _SEED.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=device)
        _SEED[key] = t
    return t


class Marks:
    """dev (bench.py --marks): named time marks inside the step, each one lane writing the device clock at that point of
    the current stream — the replayed graph's real timeline (both branches), which rocprofv3 distorts by serialising."""

    def __init__(self, device, n=256):
        self.buf = torch.zeros(n, dtype=torch.int64, device=device)
        self.names = []

    def mark(self, name):
        from . import _lib
        k = len(self.names)
        self.names.append(name)
        _lib.check(_lib.lib().gt_mark(self.buf.data_ptr() + 8 * k, _lib.current_stream(self.buf.device)), "gt_mark")

    def report(self):
        t = self.buf[:len(self.names)].cpu().tolist()
        t0 = min(v for v in t if v) if any(t) else 0
        return sorted(((v - t0) / 100.0, n) for v, n in zip(t, self.names))           # microseconds since the first mark


MARKS = None


def mark(name):
    if MARKS is not None:
        MARKS.mark(name)


def bump_seed(device):
    seed_word(device).add_(0x632BE5AB)           # odd constant: full-period walk over 2^32


# ---- per-step arena of small zero-initialised accumulators --------------------------------------------
# The backward needs ~150 tiny zeroed buffers per step (atomicAdd targets: LayerNorm gamma/beta, ActNorm
# logs/bias, InvConv dW, relative-position embeddings ...): one fill of the arena at step start instead
# of one fill launch each.  Opt-in (train.Trainer): outside a step `zeros_small` is plain torch.zeros.
_ARENA = {}
ARENA_BYTES = 1 << 20


def arena_begin(device):
    """Zero the arena and hand out slices of it until the next arena_begin (call once per training step,
    after the previous step's optimizer has consumed its gradients)."""
    key = str(device)
    a = _ARENA.get(key)
    if a is None:
        a = {"buf": torch.zeros(ARENA_BYTES, dtype=torch.uint8, device=device), "off": 0, "on": True}
        _ARENA[key] = a
    else:
        a["buf"].zero_()
    a["off"], a["on"] = 0, True


def step_head(device, extra=None, bump=True):
    """arena_begin + big_begin (+ one more region to clear: `extra` = a contiguous tensor whose byte size is a multiple of 16) and
    bump_seed in ONE launch (gt_step_zero) — the serial head of a training step; falls back to the separate calls off the GPU."""
    if device.type != "cuda":
        if bump:
            bump_seed(device)
        arena_begin(device); big_begin(device)
        if extra is not None:
            extra.zero_()
        return
    import ctypes
    key = str(device)
    regions = []
    a = _ARENA.get(key)
    if a is None:
        arena_begin(device)                            # first use: allocates (zeroed)
    else:
        regions.append((a["buf"].data_ptr(), ARENA_BYTES))
        a["off"], a["on"] = 0, True
    b = _BIG.get(key)
    if b is None:
        big_begin(device)
    else:
        n = min(BIG_BYTES, (b["used"] + 4095) & ~4095) if b["used"] else BIG_BYTES
        regions.append((b["buf"].data_ptr(), n))       # only what the previous step dirtied
        b["off"], b["on"] = 0, True
    if extra is not None and extra.numel():
        nb = extra.numel() * extra.element_size()
        if nb % 16 or extra.data_ptr() % 16 or not extra.is_contiguous():
            extra.zero_()
        else:
            regions.append((extra.data_ptr(), nb))
    args = _lib.StepZeroArgs()
    for i, (p, n) in enumerate(regions):
        args.ptr[i], args.bytes[i] = p, n
    args.n = len(regions)
    if bump:
        args.seed_word, args.seed_inc = seed_word(device).data_ptr(), 0x632BE5AB
    _lib.check(_lib.lib().gt_step_zero(ctypes.byref(args), _lib.current_stream(device)), "gt_step_zero")


def arena_end(device):
    a = _ARENA.get(str(device))
    if a is not None:
        a["on"] = False
    b = _BIG.get(str(device))
    if b is not None:
        b["on"] = False


# A second, larger pre-zeroed region for the per-layer buffers that kernels only partly write (attention outputs and their
# gradients: halo rows must read as zero): ONE fill per training step instead of one per buffer (18 launches at cfg 2).
_BIG = {}
BIG_BYTES = 96 << 20


def big_begin(device):
    key = str(device)
    b = _BIG.get(key)
    if b is None:
        b = {"buf": torch.zeros(BIG_BYTES, dtype=torch.uint8, device=device), "off": 0, "on": True, "used": 0}
        _BIG[key] = b
    else:
        n = min(BIG_BYTES, (b["used"] + 4095) & ~4095) if b["used"] else BIG_BYTES
        b["buf"][:n].zero_()                          # only what the previous step dirtied
    b["off"], b["on"] = 0, True


def zeros_big(shape, dtype, device):
    """Zeroed tensor: a slice of the step's pre-zeroed region inside a Trainer step, torch.zeros otherwise."""
    n = 1
    for d in shape:
        n *= int(d)
    b = _BIG.get(str(device))
    nbytes = (n * torch.empty(0, dtype=dtype).element_size() + 255) & ~255
    if b is None or not b["on"] or b["off"] + nbytes > BIG_BYTES:
        return torch.zeros(shape, dtype=dtype, device=device)
    off = b["off"]
    b["off"] = off + nbytes
    b["used"] = max(b["used"], b["off"])
    return b["buf"][off:off + nbytes].view(dtype)[:n].view(shape)


def zeros_small(shape, dtype, device):
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    a = _ARENA.get(str(device))
    nbytes = (n * torch.empty(0, dtype=dtype).element_size() + 255) & ~255
    if a is None or not a["on"] or nbytes > 65536 or a["off"] + nbytes > ARENA_BYTES:
        return torch.zeros(shape, dtype=dtype, device=device)
    off = a["off"]
    a["off"] = off + nbytes
    return a["buf"][off:off + nbytes].view(dtype)[:n].view(shape)


class RowsConfig:
    """Rows-layout state of ONE model (a FlowGenerator shares one object with its encoder and decoder; a stand-alone
    module uses DEFAULT_ROWS = uniform layout).  Nothing here is process-global: two trainers, or an evaluation model
    beside a trainer, do not see each other's settings."""

    def __init__(self, ragged=False, row_round=128):
        self.ragged = ragged        # utterances packed back to back (see RowsCtx); train.Trainer turns it on for its model
        self.row_round = row_round  # ragged R is rounded up to this (row tiles of the GEMMs; 512 under HIP graphs)
        self.host_lengths = {}      # "x" / "y" -> list of ints for the batch in flight (set by FlowGenerator.forward)
        self.prebuilt = {}          # "x" / "y" -> ragged RowsCtx the next forward must use (train.Trainer, around capture / replay)
        self.stamps = None          # KernelStamps for the decoder's fused WaveNet kernels (bench.py), attached to the "y" context
        # how the step is laid out over streams (defaults from the process environment, read when the model is built; the owner of
        # the model — a Trainer, a test — changes THIS object, not a module global):
        import os
        self.encoder_stream = os.environ.get("GT_ENC_STREAM", "1") != "0"      # text encoder + duration predictor as a parallel branch
        self.predictor_branch = os.environ.get("GT_PRED_BRANCH", "1") != "0"   # cfg 5: stochastic predictors on the encoder's stream
        self.energy_on_main = os.environ.get("GT_ENERGY_MAIN", "1") != "0"     # ... except the energy predictor (main stream)
        # the caller's stream waits for the predictors' branch at the end of forward() (so that the losses can be read there).
        # train.Trainer's single-backward step turns it off and seeds the backward with one root per stream instead: the decoder's
        # backward then starts behind the likelihood terms, beside the predictors' forward, instead of behind it
        self.dp_balance = os.environ.get("GT_DP_BALANCE", "1") != "0"        # long texts: the duration predictor on the caller's stream (text_models)
        self.join_predictors = True
        self.front_stream = os.environ.get("GT_FRONT_STREAM", "1") != "0"    # the conditioning front end (and its backward) on a stream of its own


DEFAULT_ROWS = RowsConfig()


def rows_add_cond(rc, x, xb, cond, want_f32=True, want_bf16=True):
    """(x or xb)[m,:] + cond[utterance(m),:] on the valid rows (gt_rows_add_cond): the speaker vector that
    attentions.py:66-67 / models.py:587-589 broadcast over time.  Returns (fp32 rows or None, bf16 rows or None)."""
    L = _lib.lib()
    src = x if x is not None else xb
    R, C = src.shape
    dev = src.device
    cond = cond.detach().float().contiguous()
    assert cond.shape == (rc.B, C), (cond.shape, rc.B, C)
    out = torch.empty(R, C, dtype=torch.float32, device=dev) if want_f32 else None
    outb = torch.empty(R, C, dtype=torch.bfloat16, device=dev) if want_bf16 else None
    _lib.check(L.gt_rows_add_cond(_lib.ptr(x), 0 if x is None else x.stride(0), _lib.ptr(xb), 0 if xb is None else xb.stride(0),
                                  _lib.ptr(cond), _lib.ptr(rc.rowmask), _lib.ptr(out), C, _lib.ptr(outb), C,
                                  rc.B, R, C, rc.Tp, _lib.ptr(rc.row0), _lib.current_stream(dev)), "gt_rows_add_cond")
    return out, outb


def cond_grad(rc, *row_grads):
    """Gradient of the vector added by rows_add_cond: per-utterance sum over the valid rows of the output gradients."""
    tot = None
    for d in row_grads:
        if d is None:
            continue
        if tot is None:
            tot = torch.empty(rc.B, d.shape[1], dtype=torch.float32, device=d.device)
            rc.utt_sum(d, tot)
        else:
            rc.utt_sum(d, tot, accumulate=True)
    return tot


def grad_accumulator(param, shape=None):
    """Zero-initialised fp32 buffer that a backward kernel accumulates a (non-conv) parameter's gradient into with
    atomics.  Under train.Trainer it is the parameter's own slice of the flat gradient buffer (train.GradBuckets lays
    these parameters out first and zeroes the region once per step), so no copy is needed afterwards; elsewhere it is
    a zeros_small buffer."""
    shape = tuple(shape) if shape is not None else tuple(param.shape)
    fg = getattr(param, "_gt_flat_grad", None)
    gb = getattr(param, "_gt_bucket", None)      # train.GradBuckets: live between its zero_accum() and the step's last gather()
    if fg is not None and gb is not None and gb.accum_live and getattr(param, "_gt_prezeroed", False):
        buf, off = fg
        return buf[off:off + param.numel()].view(shape)
    return zeros_small(shape, torch.float32, param.device)


class RowsCtx:
    """Geometry of one batch in the rows layout; rowmask is 1 on valid frames.

    uniform (default): utterance b owns rows [b*Tp, (b+1)*Tp), Tp = T + 2*HALO, frame t is row b*Tp + HALO + t.
    ragged  (lengths_host given): utterance b owns rows [row0[b], row0[b+1]) = its OWN frames + 2*HALO, back to
    back — padded frames (30 % of an LJSpeech-shaped batch) cost no GEMM, no elementwise and no HBM work.  R is
    rounded up to `round_to` (the last utterance owns the extra, masked rows) so that a captured HIP graph can be
    replayed for every batch of the same rounded size; Tp = the largest row count of one utterance (grid sizing).
    The host needs the lengths (the data loader has them; no device sync)."""

    stamps = None           # ops.KernelStamps or None (set by make_ctx for the decoder's context)

    def __init__(self, lengths, T, lengths_host=None, round_to=None):
        _lib.require_cuda(lengths)
        self.device = lengths.device
        self.B = int(lengths.shape[0])
        self.T = int(T)
        self.lengths = lengths.to(torch.int32)
        self.ragged = lengths_host is not None
        if not self.ragged:
            self.Tp = self.T + 2 * HALO
            self.R = self.B * self.Tp
            self.row0 = None
            t = torch.arange(self.Tp, device=self.device) - HALO
            self.rowmask2d = ((t[None, :] >= 0) & (t[None, :] < self.lengths[:, None])).to(torch.float32)
            self.rowmask = self.rowmask2d.reshape(-1).contiguous()
            self.rowutt = (torch.arange(self.R, device=self.device) // self.Tp).to(torch.int32)
            return
        assert len(lengths_host) == self.B
        rnd = self.rnd = int(round_to or DEFAULT_ROWS.row_round)
        starts, self.R = self.row_starts(lengths_host, self.T, rnd)  # the last utterance owns the rounding rows
        # Tp only sizes grids: the upper bound keeps a captured graph valid for any batch with the same R
        self.Tp = self.T + 2 * HALO + rnd - 1
        self.rowmask2d = None
        assert not torch.cuda.is_current_stream_capturing(), \
            "a ragged RowsCtx is built outside graph capture (train.Trainer prebuilds and refreshes it)"
        # row0 [B+1] and lengths [B] share ONE device buffer: a refresh is a single host-to-device copy of both
        self._geo = torch.empty(2 * self.B + 1, dtype=torch.int32, device=self.device)
        self.row0, self.lengths = self._geo[:self.B + 1], self._geo[self.B + 1:]
        self._ring, self._ring_i = [], 0                            # pinned host staging buffers of the refreshes in flight
        self.rowbatch = torch.empty(self.R, dtype=torch.int64, device=self.device)
        self.rowframe = torch.empty(self.R, dtype=torch.int32, device=self.device)
        self.rowmask = torch.empty(self.R, dtype=torch.float32, device=self.device)
        self.rowutt = torch.empty(self.R, dtype=torch.int32, device=self.device)
        self._fill(starts, lengths_host)

    RING = 8

    def _stage(self, starts, lengths_host):
        """row0 / lengths (host lists) into the next pinned staging buffer of the ring -> (buffer, its event: to be recorded behind
        the launch that reads the buffer)."""
        if len(self._ring) < self.RING:
            self._ring.append((torch.empty(2 * self.B + 1, dtype=torch.int32).pin_memory(), torch.cuda.Event()))
            host, ev = self._ring[-1]
        else:
            host, ev = self._ring[self._ring_i % self.RING]
            ev.synchronize()                                         # the launch that last read this buffer (8 refreshes ago) is done
        self._ring_i += 1
        hv = host.numpy()
        hv[:self.B + 1] = starts
        hv[self.B + 1:] = [int(v) for v in lengths_host]
        return host, ev

    def stage_refresh(self, lengths_host):
        """refresh() in two halves, for a caller that launches gt_step_inputs itself (train.Trainer: the batch's copies and every
        context in one launch): -> None if the rounded size differs, else (job fields for _lib.StepCtx, pinned buffer, event)."""
        starts, R = self.row_starts(lengths_host, self.T, self.rnd)
        if R != self.R or self.B > _lib.STEP_MAX_B:
            return None
        host, ev = self._stage(starts, lengths_host)
        job = dict(geo_src=host.data_ptr(), geo_dst=self._geo.data_ptr(), rowbatch=self.rowbatch.data_ptr(), rowframe=self.rowframe.data_ptr(),
                   rowmask=self.rowmask.data_ptr(), rowutt=self.rowutt.data_ptr(), B=self.B, R=self.R)
        return job, host, ev

    def _fill(self, starts, lengths_host):
        """row0 / lengths (host lists) -> device (ONE non-blocking copy from pinned memory), then row -> utterance, row ->
        frame, rowmask in ONE launch; all in place, stream-ordered."""
        host, ev = self._stage(starts, lengths_host)
        self._geo.copy_(host, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.device))
        _lib.check(_lib.lib().gt_rows_ctx_fill(_lib.ptr(self.row0), _lib.ptr(self.lengths), _lib.ptr(self.rowbatch),
                                               _lib.ptr(self.rowframe), _lib.ptr(self.rowmask), _lib.ptr(self.rowutt), self.B, self.R,
                                               _lib.current_stream(self.device)), "gt_rows_ctx_fill")

    def row_utt(self):
        """int32 [R]: utterance of every row (rows past the last utterance's frames belong to the last one)."""
        return self.rowutt

    def utt_sum(self, rows, out, accumulate=False, masked=True):
        """out[b, :] (+)= sum of the (valid) rows of utterance b; rows bf16 or fp32 [R, C] (a column slice is fine),
        out fp32 [B, C] (a column slice is fine) — gt_rows_utt_sum."""
        L = _lib.lib()
        C = rows.shape[1]
        assert rows.stride(1) == 1 and out.stride(1) == 1 and out.shape == (self.B, C) and out.dtype == torch.float32
        _lib.check(L.gt_rows_utt_sum(_lib.ptr(rows), rows.stride(0), int(rows.dtype == torch.float32),
                                     _lib.ptr(self.rowmask) if masked else None, _lib.ptr(out), out.stride(0), int(accumulate),
                                     self.B, self.R, C, self.Tp, _lib.ptr(self.row0), _lib.current_stream(rows.device)),
                   "gt_rows_utt_sum")
        return out

    @staticmethod
    def row_starts(lengths_host, T, rnd):
        """(starts [B+1] with starts[B] = R rounded up, R) of the ragged layout for these lengths."""
        starts = [0]
        for v in lengths_host:
            starts.append(starts[-1] + max(0, min(int(v), T)) + 2 * HALO)
        R = -(-starts[-1] // rnd) * rnd
        starts[-1] = R
        return starts, R

    def refresh(self, lengths, lengths_host):
        """New batch of the same rounded size (a captured graph that reads this context is about to be replayed):
        stream-ordered, in-place update of everything the graph's kernels read.  False if the size differs."""
        starts, R = self.row_starts(lengths_host, self.T, self.rnd)
        if R != self.R:
            return False
        self._fill(starts, lengths_host)                # the host lengths ARE the lengths (the device copy is not read back)
        return True

    def mask_bt(self):
        """[B, T] 1/0 mask of valid frames."""
        return (torch.arange(self.T, device=self.device)[None, :] < self.lengths[:, None]).to(torch.float32)

    def to_rows(self, x, dtype=None):
        """[B, C, T] -> [R, C] (zero halos and padding rows; frames past an utterance's length are copied as they are — callers
        mask) — one launch (gt_rows_from_bct) for fp32 / bf16 tensors."""
        B, C, T = x.shape
        assert B == self.B and T == self.T
        dtype = dtype or x.dtype
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and dtype in (torch.float32, torch.bfloat16):
            xc = x.contiguous()
            out = torch.empty(self.R, C, device=x.device, dtype=dtype)
            _lib.check(_lib.lib().gt_rows_from_bct(_lib.ptr(xc), int(xc.dtype == torch.float32), _lib.ptr(out), int(dtype == torch.float32),
                                                   _lib.ptr(self.row0) if self.ragged else None, B, C, T, self.Tp, self.R,
                                                   _lib.current_stream(x.device)), "gt_rows_from_bct")
            return out
        if not self.ragged:
            out = torch.zeros(self.B, self.Tp, C, device=x.device, dtype=dtype)
            out[:, HALO:HALO + T] = x.transpose(1, 2).to(out.dtype)
            return out.reshape(self.R, C)
        xt = x.transpose(1, 2).to(dtype)                                             # [B, T, C]
        inside = (self.rowframe >= 0) & (self.rowframe < T)
        idx = self.rowbatch.long() * T + self.rowframe.clamp(0, T - 1).long()
        return (xt.reshape(B * T, C)[idx] * inside[:, None].to(xt.dtype)).contiguous()

    def from_rows(self, xr, dtype=None):
        """[R, C] -> [B, C, T] (frames that have no row — past an utterance's length — come back as zero) — one launch
        (gt_bct_from_rows) for fp32 / bf16 rows."""
        C = xr.shape[1]
        dtype = dtype or xr.dtype
        if xr.is_cuda and xr.dtype in (torch.float32, torch.bfloat16) and dtype in (torch.float32, torch.bfloat16) and self.ragged:
            rows = xr.contiguous()
            out = torch.empty(self.B, C, self.T, device=xr.device, dtype=dtype)
            _lib.check(_lib.lib().gt_bct_from_rows(_lib.ptr(rows), int(rows.dtype == torch.float32), _lib.ptr(out), int(dtype == torch.float32),
                                                   _lib.ptr(self.lengths), _lib.ptr(self.row0), self.B, C, self.T, self.Tp, self.R,
                                                   _lib.current_stream(xr.device)), "gt_bct_from_rows")
            return out
        if not self.ragged:
            x = xr.reshape(self.B, self.Tp, C)[:, HALO:HALO + self.T].transpose(1, 2)
            return x.to(dtype).contiguous()
        t = torch.arange(self.T, device=self.device)
        rows = (self.row0[:-1].long()[:, None] + HALO + t[None, :]).clamp_(max=self.R - 1)   # [B, T]
        out = xr[rows.reshape(-1)].reshape(self.B, self.T, C) * self.mask_bt()[:, :, None].to(xr.dtype)
        return out.transpose(1, 2).to(dtype).contiguous()


def length_mask(lengths, T, dtype=torch.float32):
    """[B, 1, T] mask of the frames t < lengths[b] (commons.sequence_mask(...).unsqueeze(1)) in one launch."""
    B = lengths.shape[0]
    if not lengths.is_cuda or dtype != torch.float32:
        return (torch.arange(T, device=lengths.device)[None, :] < lengths[:, None]).unsqueeze(1).to(dtype)
    out = torch.empty(B, 1, T, dtype=torch.float32, device=lengths.device)
    ln = lengths if lengths.dtype in (torch.int32, torch.int64) else lengths.long()
    _lib.check(_lib.lib().gt_length_mask(_lib.ptr(ln.contiguous()), int(ln.dtype == torch.int64), _lib.ptr(out), B, T,
                                         _lib.current_stream(lengths.device)), "gt_length_mask")
    return out


def make_ctx(lengths, T, which, div=1, cfg=None):
    """RowsCtx for the text side (which = "x") or the mel side ("y", div = n_sqz) under the model's RowsConfig: ragged
    when cfg.ragged is on and the host knows the lengths of the batch in flight (FlowGenerator.forward / train.Trainer
    put them in), else uniform."""
    cfg = cfg or DEFAULT_ROWS
    pre = cfg.prebuilt.get(which)
    if pre is not None:                                     # the captured-graph path: context built (and refreshed) outside
        assert pre.T == int(T), "prebuilt rows context does not fit this batch"
        rc = pre
    else:
        if callable(lengths):                               # the lengths tensor costs launches: only computed when it is needed
            lengths = lengths()
        lh = cfg.host_lengths.get(which) if cfg.ragged else None
        rc = RowsCtx(lengths, T) if lh is None else RowsCtx(lengths, T, lengths_host=[int(v) // div for v in lh], round_to=cfg.row_round)
    if which == "y":
        rc.stamps = cfg.stamps
    return rc


class PackSlice:
    """Pack destination that is a window of a bigger packed image: rows [k0, k0+Cin) of the reduction axis of
    a K-concatenated GEMM (`parent`), same attribute names as PackedConv for the pack descriptors."""

    def __init__(self, parent, k0, Cout, Cin):
        assert parent.taps == 1 and not parent.gate
        self.Cout, self.Cin, self.taps, self.gate = Cout, Cin, 1, False
        self.Np_f, self.Kp_f, self.Np_d, self.Kp_d = parent.Np_f, parent.Kp_f, parent.Np_d, parent.Kp_d
        # forward image Pf[co][k0 + ci] / data-gradient image Pd[k0 + ci][co]
        assert k0 % 32 == 0
        if getattr(parent, "frag", False):          # fragment order: [n / 32][k / 16] 1-KB (512-element) fragments
            self.fwd = parent.fwd[(k0 >> 4) * 512:]
            self.dgrad = parent.dgrad[(k0 >> 5) * (parent.Kp_d >> 4) * 512:]
            self.flags = 6
        else:
            self.fwd = parent.fwd[k0:]
            self.dgrad = parent.dgrad[k0 * parent.Kp_d:]
            self.flags = 0
        self.inv_norm = None


class PackSliceN:
    """The transposed window: output channels [n0, n0+Cout) of an N-concatenated GEMM (`parent`, taps = 1) — q, k and v
    projections of one attention layer as ONE [R, C] x [C, 3C] GEMM forward and one K = 3C GEMM for the data gradient."""

    def __init__(self, parent, n0, Cout, Cin):
        assert parent.taps == 1 and not parent.gate and n0 % 32 == 0
        self.Cout, self.Cin, self.taps, self.gate = Cout, Cin, 1, False
        self.Np_f, self.Kp_f, self.Np_d, self.Kp_d = parent.Np_f, parent.Kp_f, parent.Np_d, parent.Kp_d
        # forward image Pf[n0 + co][ci] / data-gradient image Pd[ci][n0 + co]
        self.fwd = parent.fwd[n0 * parent.Kp_f:]
        self.dgrad = parent.dgrad[n0:]
        self.inv_norm = None
        self.flags = 0


class PackedConv:
    """bf16 MFMA-ready images of one conv's weight: forward and data-gradient packing (row-major [taps][Np][Kp])."""

    def __init__(self, Cout, Cin, taps, gate=False, device="cuda", norm_only=False, split3=False, frag=False, gate16=False):
        """norm_only: only the per-row 1/||v|| is wanted (the weight itself is packed elsewhere, e.g. into the
        K-concatenated skip GEMM of a WN): no bf16 images.
        split3: bf16x3 images ([w_hi; w_lo; w_hi] along the reduction axis, gt_pack_conv_weights flag 8) for a near-fp32
        product against [x_hi | x_hi | x_lo] activations — the stochastic predictors' 1x1 convs."""
        self.Cout, self.Cin, self.taps, self.gate = Cout, Cin, taps, gate
        self.split3 = bool(split3)
        self.km = 3 if split3 else 1                # reduction-axis multiplier of the packed images
        # frag: both images in MFMA-fragment order (flags 2 | 4: one 1-KB A-fragment per (32 channels, 16 k)); gate16: the
        # gate interleave at MFMA-block granularity (flag 16) — what the fused WaveNet-layer kernels read (csrc/wn_layer.hip)
        self.frag, self.gate16 = bool(frag), bool(gate16)
        assert not (frag and split3) and not (gate16 and not gate)
        self.inv_norm = torch.zeros(Cout, dtype=torch.float32, device=device)
        self.fwd = self.dgrad = None
        self.Kp_f = self.Np_f = self.Kp_d = self.Np_d = 0
        if norm_only:
            return
        assert not (split3 and (gate or taps != 1))
        self.Kp_f = _round_up(self.km * Cin, 64)
        self.Np_f = Cout if gate else (_round_up(Cout, 128) if Cout % 128 == 0 else _round_up(Cout, 64))
        self.Kp_d = _round_up(self.km * Cout, 64)
        self.Np_d = _round_up(Cin, 128) if _round_up(Cin, 64) % 128 == 0 else _round_up(Cin, 64)
        self.fwd = torch.zeros(taps * self.Np_f * self.Kp_f, dtype=torch.int16, device=device)
        self.dgrad = torch.zeros(taps * self.Np_d * self.Kp_d, dtype=torch.int16, device=device)

    @property
    def flags(self):
        g = 16 if (self.gate and self.gate16) else int(bool(self.gate))
        return g + 6 * int(self.frag) + 8 * int(self.split3)

    def pack(self, v, g=None):
        """v: [Cout, Cin, taps] fp32 (weight_v or plain weight), g: [Cout,1,1] or None."""
        L = _lib.lib()
        v = v.detach().contiguous().float()
        assert tuple(v.shape) == (self.Cout, self.Cin, self.taps), (v.shape, self.Cout, self.Cin, self.taps)
        gg = None if g is None else g.detach().reshape(-1).contiguous().float()
        _lib.check(L.gt_pack_conv_weights(_lib.ptr(v), _lib.ptr(gg), _lib.ptr(self.fwd), _lib.ptr(self.dgrad),
                                          _lib.ptr(self.inv_norm), self.Cout, self.Cin, self.taps,
                                          self.Np_f, self.Kp_f, self.Np_d, self.Kp_d, self.flags,
                                          _lib.current_stream(v.device)), "gt_pack_conv_weights")
        return self


def conv_rows(x, pc, ctx, *, dgrad=False, bias=None, cond=None, mask=False, out=None, out_f32=False,
              addend=None, relu=False, gate=False, gate_t=None, gate_s=None, drop_p=0.0, seed=0, R=None, tag=None,
              cond_per_row=False, tile=0):
    """Y = epilogue(conv(x)) in the rows layout via gt_conv_gemm_bf16.  x: [R, >=Cin] bf16.
    `out`/`addend` may be column-slices of wider row buffers (row stride taken from .stride(0)).
    tile: _lib.GT_TILE_* (0 = the library's choice; tests force every variant)."""
    L = _lib.lib()
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    R = x.shape[0] if R is None else R
    km = getattr(pc, "km", 1)                       # bf16x3 split images: x is [hi | hi | lo], three times as wide
    if dgrad:
        N, Cin, Np, Kp, W = pc.Cin, km * pc.Cout, pc.Np_d, pc.Kp_d, pc.dgrad
    else:
        N, Cin, Np, Kp, W = pc.Cout, km * pc.Cin, pc.Np_f, pc.Kp_f, pc.fwd
    assert x.shape[1] >= Cin, (x.shape, Cin)
    n_out = N // 2 if gate is True or gate == 1 else (2 * N if gate == 2 else N)      # gate == 3: ReLU / dropout backward from the saved y (gate_t)
    if out is None:
        out = torch.empty(R, n_out, device=x.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    out_f32 = out.dtype == torch.float32
    if gate == 1:
        if gate_t is None:
            gate_t = torch.empty(R, n_out, device=x.device, dtype=torch.bfloat16)
            gate_s = torch.empty(R, n_out, device=x.device, dtype=torch.bfloat16)
    _ev = KERNEL_TIMER.start(tag)
    rc = L.gt_conv_gemm_bf16(_lib.ptr(x), x.stride(0), _lib.ptr(W), _lib.ptr(bias),
                             _lib.ptr(cond), 0 if cond is None else cond.stride(0),
                             _lib.ptr(ctx.rowmask) if mask else None,
                             _lib.ptr(out), out.stride(0), int(out_f32),
                             _lib.ptr(addend), 0 if addend is None else addend.stride(0),
                             _lib.ptr(gate_t), _lib.ptr(gate_s), 0 if gate_t is None else gate_t.stride(0),
                             R, N, Cin, pc.taps, ctx.Tp, Np, Kp, int(relu), int(gate), float(drop_p), int(seed),
                             _lib.ptr(seed_word(x.device)) if (drop_p > 0 and gate != 3) else None,
                             _lib.ptr(ctx.row0) if (cond is not None and not cond_per_row) else None,
                             0 if cond_per_row else ctx.B, int(tile), _lib.current_stream(x.device))
    KERNEL_TIMER.stop(_ev)
    _lib.check(rc, "gt_conv_gemm_bf16")
    return (out, gate_t, gate_s) if gate == 1 else out
