"""Minimal data-parallel training harness for the hot path (the call pattern of reference
train_ms_emo_lang_pitch.py:281-314 / train.py:112-150 on synthetic batches): zero_grad, forward,
mle + duration loss, backward, gradient all-reduce (RCCL over xGMI via torch.distributed), grad
norm, optimizer step.

One process per GPU.  Gradients are packed into a few large flat buckets (sized for xGMI: few,
large collectives) and all-reduced on a side stream as soon as the backward of the decoder /
encoder that produced them has finished, overlapping with the rest of the backward pass.
"""
import json
import math
import os

import contextlib
import torch
import torch.distributed as dist

from . import _lib, models

BASE_MODEL = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1,
                  n_blocks_dec=12, n_layers_enc=6, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5,
                  n_block_layers=4, n_sqz=2, prenet=True, mean_only=True, hidden_channels_enc=192,
                  hidden_channels_dec=192, window_size=4,      # == reference configs/base.json "model"
                  use_sdp=False)   # the upstream-equivalent live sub-graph (deterministic DurationPredictor; SURVEY F1: the fork's
                                   # class, whose default is use_sdp=True, does not construct for the base configs)


def load_model_config(path=None):
    """Model kwargs from a reference-style JSON config (configs/base.json layout), else base.json's values."""
    if path and os.path.exists(path):
        with open(path) as f:
            return dict(json.load(f)["model"])
    return dict(BASE_MODEL)


def build_model(cfg=None, n_vocab=148, out_channels=80, device="cuda"):
    cfg = dict(cfg or BASE_MODEL)
    return models.FlowGenerator(n_vocab=n_vocab, out_channels=out_channels, **cfg).to(device)


def synth_batch(B, Tx_max, Ty_max, rank, device, n_vocab=148, blank=False):
    """Synthetic LJSpeech-shaped batch: seed 1234 + rank, T_x ~ U{60..150},
    T_y ~ 2*U{150..400}, one sample pinned at the maximum, mel ~ N(0,1) masked, ids ~ U{1..n_vocab-1}."""
    g = torch.Generator().manual_seed(1234 + rank)
    lo_x = max(1, int(Tx_max * 0.4))
    t_x = torch.randint(lo_x, Tx_max + 1, (B,), generator=g)
    t_y = torch.randint(max(1, Ty_max * 3 // 16), Ty_max // 2 + 1, (B,), generator=g) * 2
    t_y = torch.maximum(t_y, t_x + (t_x % 2))
    t_x[0], t_y[0] = Tx_max, Ty_max
    ids = torch.randint(1, n_vocab, (B, Tx_max), generator=g)
    ids = ids * (torch.arange(Tx_max)[None, :] < t_x[:, None])
    y = torch.randn(B, 80, Ty_max, generator=g) * (torch.arange(Ty_max)[None, None, :] < t_y[:, None, None])
    return ids.to(device), t_x.to(device), y.to(device), t_y.to(device)


def one_cycle(step, total_steps, max_lr, pct_start=0.3, div_factor=25.0, final_div_factor=1e4,
              base_momentum=0.85, max_momentum=0.95):
    """(lr, beta1) of torch.optim.lr_scheduler.OneCycleLR with its defaults (cosine, two phases, momentum
    cycled inversely to the LR) — the schedule of reference train_ms_emo_lang_pitch.py:161,314."""
    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    initial, min_lr = max_lr / div_factor, max_lr / div_factor / final_div_factor
    end1 = float(pct_start * total_steps) - 1.0
    end2 = float(total_steps) - 1.0
    step = min(float(step), end2)
    if step <= end1:
        pct = step / end1 if end1 > 0 else 1.0
        return cos(initial, max_lr, pct), cos(max_momentum, base_momentum, pct)
    pct = (step - end1) / (end2 - end1)
    return cos(max_lr, min_lr, pct), cos(base_momentum, max_momentum, pct)


class GradBuckets:
    """All gradients in ONE flat fp32 buffer + asynchronous all-reduce (mean) of large slices of it on a
    communication stream.  Every parameter gets `_gt_flat_grad = (buffer, offset)`: the deferred weight-gradient
    kernels (wgrad.WgradQueue) then write their results straight into the buffer, the others are copied in by
    `gather()` (one multi-tensor copy); the optimizer and the collectives only ever see the flat buffer."""
    ALIGN = 64          # floats: every parameter starts on a 256-byte boundary

    def __init__(self, params, world, bucket_mb=64, accum=(), wire="fp32", force_collectives=False):
        """wire: "fp32" — all-reduce (mean) of the fp32 buffer, what the reference's DDP does; "bf16" — half the bytes on
        xGMI with fp32 accumulation: every rank sends bf16 copies of the other ranks' shards (all-to-all), sums the copies of
        its own shard in fp32, and the reduced shards travel back as bf16 (all-gather); all ranks end up with bit-identical
        gradients, rounded to bf16 twice (opt-in: `Trainer(grad_wire="bf16")`).
        accum: parameters whose gradients are ACCUMULATED by atomics (LayerNorm, ActNorm, InvConvNear, relative-
        position and token embeddings): they are laid out first, contiguously, and `zero_accum()` clears that region
        with one fill per step, so that their backward kernels add straight into the flat buffer
        (ops.grad_accumulator) — every other gradient is overwritten whole by the batched wgrad kernels."""
        self.world = world
        # force_collectives: issue the step's collectives even with one rank (a world-1 process group) — how the RCCL path (ReduceOp.AVG,
        # all-to-all / all-gather of the bf16 wire, the communication stream's hand-offs) is exercised on a one-GPU box
        self.collect = world > 1 or bool(force_collectives)
        assert wire in ("fp32", "bf16")
        self.wire, self._wire_bufs = wire, {}
        ids = {id(p) for p in accum}
        ps = [p for p in params if p.requires_grad]
        self.params = [p for p in ps if id(p) in ids] + [p for p in ps if id(p) not in ids]
        self.n_accum = sum(1 for p in ps if id(p) in ids)
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.total = off
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        cap = max(self.ALIGN, int(bucket_mb * (1 << 20) / 4) // self.ALIGN * self.ALIGN)
        self.buckets = [(s, min(s + cap, self.total)) for s in range(0, self.total, cap)]
        self.accum_live = False       # True between zero_accum() and the last gather() of a step (ops.grad_accumulator)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            p._gt_flat_grad = (self.flat, o)
            p._gt_bucket = self
            p._gt_prezeroed = i < self.n_accum
        self.accum_end = self.offsets[self.n_accum] if self.n_accum < len(self.params) else self.total
        self.on_gpu = dev.type == "cuda"
        self.comm = torch.cuda.Stream(device=dev) if (self.collect and self.on_gpu) else None
        self.active = [True] * len(self.params)

    def zero_accum(self):
        """Start of a step: clear the accumulated-gradient region and let ops.grad_accumulator hand out its slices
        (until the step's last gather(); a model run outside a Trainer step never sees the flat buffer)."""
        if self.n_accum:
            self.flat[:self.accum_end].zero_()
            self.accum_live = True

    def accum_region(self):
        """zero_accum() for a caller that clears the region itself (ops.step_head: one launch for all of a step's fills): -> the slice
        to clear (None if there is none); ops.grad_accumulator hands out its slices from now on."""
        if not self.n_accum:
            return None
        self.accum_live = True
        return self.flat[:self.accum_end]

    def view(self, i):
        p, o = self.params[i], self.offsets[i]
        return self.flat[o:o + p.numel()].view_as(p)

    def gather(self, lo=0, hi=None):
        """Make p.grad the parameter's slice of the flat buffer for parameters [lo, hi) (copy only what is not there
        already)."""
        from . import wgrad
        wgrad.join(self.flat.device)             # weight gradients flushed on the side stream land first
        dsts, srcs = [], []
        hi = len(self.params) if hi is None else hi
        if lo == 0:
            self.accum_live = False
        for i in range(lo, hi):
            p = self.params[i]
            g = p.grad
            self.active[i] = g is not None
            if g is None:
                continue
            v = self.view(i)
            if g.data_ptr() != v.data_ptr():
                dsts.append(v); srcs.append(g)
            p.grad = v
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def active_runs(self):
        """[(start, end)] float ranges of the flat buffer covering runs of parameters that have a gradient
        (torch optimizers skip parameters whose grad is None; so does the flat AdamW)."""
        runs, start = [], None
        for i, a in enumerate(self.active):
            if a and start is None:
                start = self.offsets[i]
            if not a and start is not None:
                runs.append((start, self.offsets[i])); start = None
        if start is not None:
            runs.append((start, self.total))
        return runs

    def reduce_all(self):
        """gather(), then all-reduce (mean) the flat buffer slice by slice on the communication stream."""
        self.gather()
        self.allreduce()

    def allreduce(self, lo=0, hi=None, wait=True):
        """All-reduce (mean) floats [lo, hi) of the flat buffer in bucket-sized pieces.  GPU: on the communication
        stream, after everything queued so far on the current stream; wait=False leaves the current stream free to run
        ahead (the caller overlaps the backward of the remaining parameters) until `wait_comm()`."""
        hi = self.total if hi is None else hi
        if not self.collect or hi <= lo:
            return
        pieces = [(max(s, lo), min(e, hi)) for s, e in self.buckets if min(e, hi) > max(s, lo)]
        if self.wire == "bf16":
            if self.on_gpu:
                cur = torch.cuda.current_stream()
                self.comm.wait_stream(cur)
                with torch.cuda.stream(self.comm):
                    for s, e in pieces:
                        self._allreduce_bf16(s, e)
                if wait:
                    cur.wait_stream(self.comm)
            else:
                for s, e in pieces:
                    self._allreduce_bf16(s, e)
            return
        if self.on_gpu:
            cur = torch.cuda.current_stream()
            self.comm.wait_stream(cur)
            avg = dist.get_backend() == "nccl"                                  # gloo (tests) has no AVG: SUM, then scale
            with torch.cuda.stream(self.comm):
                for s, e in pieces:
                    dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)   # RCCL over xGMI
                    if not avg:
                        self.flat[s:e].mul_(1.0 / self.world)
            if wait:
                cur.wait_stream(self.comm)
        else:                                                                   # gloo (CPU tests): SUM then scale
            for s, e in pieces:
                dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM)
                self.flat[s:e].mul_(1.0 / self.world)

    def _allreduce_bf16(self, lo, hi):
        """mean of floats [lo, hi) over the ranks with bf16 on the wire and fp32 accumulation (see __init__)."""
        w, rank, n = self.world, dist.get_rank(), hi - lo
        shard = (-(-n // w) + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        bufs = self._wire_bufs.get((lo, hi))
        if bufs is None:
            dev = self.flat.device
            bufs = self._wire_bufs[(lo, hi)] = tuple(torch.zeros(w * shard, dtype=torch.bfloat16, device=dev) for _ in range(3)) + \
                (torch.empty(shard, dtype=torch.bfloat16, device=dev),)
        send, recv, full, mine = bufs
        send[:n].copy_(self.flat[lo:hi])                                        # fp32 -> bf16; the padding stays zero
        if dist.get_backend() == "nccl":
            dist.all_to_all_single(recv, send)                                  # recv[j] = rank j's copy of MY shard
            copies = recv.view(w, shard)
        else:                                                                   # gloo (tests) has no all-to-all
            every = [torch.empty_like(send) for _ in range(w)]
            dist.all_gather(every, send)
            copies = torch.stack([t[rank * shard:(rank + 1) * shard] for t in every])
        mine.copy_(copies.float().sum(0).mul_(1.0 / w))                         # fp32 accumulate, bf16 back onto the wire
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(full, mine)
        else:
            parts = [torch.empty_like(mine) for _ in range(w)]
            dist.all_gather(parts, mine)
            full.copy_(torch.cat(parts))
        self.flat[lo:hi].copy_(full[:n])

    def wait_comm(self):
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)


class FlatAdamW:
    """torch.optim.AdamW semantics on flat buffers through gt_adamw_flat (one HBM pass, gradient norm included).
    Parameters are re-pointed at slices of one flat fp32 buffer laid out like the GradBuckets buffer.
    Parameters whose grad is None are skipped like torch does; the bias-correction step count is global (torch
    keeps one per parameter, which only differs for a parameter that gets gradients in some steps and not others)."""

    def __init__(self, gb, lr, betas, eps, weight_decay=0.01):
        self.gb = gb
        dev = gb.flat.device
        self.flat_p = torch.zeros_like(gb.flat)
        with torch.no_grad():
            for p, o in zip(gb.params, gb.offsets):
                self.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view_as(p)
        self.m = torch.zeros_like(gb.flat)
        self.v = torch.zeros_like(gb.flat)
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay, 0.0], dtype=torch.float32, device=dev)
        self._host = torch.empty(2, dtype=torch.float32)
        if dev.type == "cuda":
            self._host = self._host.pin_memory()
        self.gnorm_sq = None

    def set_schedule(self, lr, beta1):
        """Host-side schedule update (outside a captured graph: the graph's kernels read it from device memory)."""
        self._host[0], self._host[1] = lr, beta1
        self.hyper[:2].copy_(self._host, non_blocking=True)

    _early = None            # (lo, hi): floats already updated in this step by step_early()

    def _begin(self):
        from . import ops
        self.hyper[5:6].add_(1.0)
        self.gnorm_sq = ops.zeros_small(1, torch.float32, self.flat_p.device)

    def _launch(self, s, e):
        dev = self.flat_p.device
        _lib.check(_lib.lib().gt_adamw_flat(self.flat_p.data_ptr() + 4 * s, self.gb.flat.data_ptr() + 4 * s, self.m.data_ptr() + 4 * s,
                                            self.v.data_ptr() + 4 * s, e - s, _lib.ptr(self.hyper), _lib.ptr(self.gnorm_sq),
                                            _lib.current_stream(dev)), "gt_adamw_flat")

    def step_early(self, lo, hi):
        """Update floats [lo, hi) NOW, on the current stream, before the rest of the backward has finished: the caller knows that
        every gradient in the range is final and that every parameter in it has one (train.Trainer: the decoder's conv parameters,
        90 % of the buffer, right behind the decoder's batched weight gradients — the pass then runs beside the text encoder's last
        backward launches instead of behind them).  step() later covers what is left."""
        assert self._early is None and 0 <= lo < hi <= self.gb.total
        self._begin()
        self._launch(lo, hi)
        self._early = (lo, hi)

    def step(self):
        if self._early is None:
            self._begin()
            lo = hi = -1
        else:
            lo, hi = self._early
            self._early = None
        for s, e in self.gb.active_runs():
            for a, b in ((s, min(e, lo)), (max(s, hi), e)) if hi > lo else ((s, e),):
                if b > a:
                    self._launch(a, b)
        return self.gnorm_sq


# other threads (the RCCL watchdog of torch.distributed, data loaders) may touch the runtime while this thread captures
CAPTURE_MODE = "thread_local"


def _copy_padded(dst, src):
    """dst[..., :T] = src, dst[..., T:] = 0 where T = src's (shorter or equal) last dimension: a batch goes straight into the
    padded static buffer of its graph (two launches at most; padding it first and copying the padded tensor took three)."""
    T = src.shape[-1] if src.dim() else 0
    if src.dim() == 0 or dst.shape == src.shape:
        dst.copy_(src)
        return
    assert dst.shape[:-1] == src.shape[:-1] and dst.shape[-1] >= T
    dst[..., :T].copy_(src)
    dst[..., T:].zero_()


def _upload_step_inputs(pairs, ctx_lens):
    """The batch into a captured step's static buffers — (dst, src) pairs, dst padded along the last dimension — and its ragged row
    contexts rebuilt for the batch's lengths: ONE launch (gt_step_inputs) where every tensor took a copy + a fill and every context a
    host-to-device copy + a launch; whatever does not fit the kernel's form (non-contiguous, odd sizes) goes the old way."""
    args = _lib.StepInputsArgs()
    n = 0
    for dst, src in pairs:
        if dst.data_ptr() == src.data_ptr():
            continue
        ok = (n < _lib.STEP_MAX_COPIES and src.is_cuda and dst.is_cuda and src.dtype == dst.dtype and src.is_contiguous() and dst.is_contiguous()
              and src.dim() >= 1 and dst.shape[:-1] == src.shape[:-1] and dst.shape[-1] >= src.shape[-1] and src.numel() > 0
              and (src.shape[-1] * src.element_size()) % 4 == 0 and (dst.shape[-1] * dst.element_size()) % 4 == 0
              and src.data_ptr() % 4 == 0 and dst.data_ptr() % 4 == 0)
        if not ok:
            _copy_padded(dst, src)
            continue
        c = args.copy[n]
        c.src, c.dst = src.data_ptr(), dst.data_ptr()
        c.rows = src.numel() // src.shape[-1]
        c.src_words, c.dst_words = src.shape[-1] * src.element_size() // 4, dst.shape[-1] * dst.element_size() // 4
        n += 1
    args.n_copy = n
    staged = []
    for ctx, lens in ctx_lens:
        st = ctx.stage_refresh(lens) if len(staged) < _lib.STEP_MAX_CTX else None
        if st is None:
            assert ctx.refresh(None, lens), "row count of the batch does not match the captured graph"
            continue
        job, host, ev = st
        c = args.ctx[len(staged)]
        for k, v in job.items():
            setattr(c, k, v)
        staged.append((host, ev))
    args.n_ctx = len(staged)
    if n or staged:
        import ctypes
        dev = pairs[0][0].device if pairs else ctx_lens[0][0].device
        _lib.check(_lib.lib().gt_step_inputs(ctypes.byref(args), _lib.current_stream(dev)), "gt_step_inputs")
        for host, ev in staged:
            ev.record(torch.cuda.current_stream(dev))


def graph_key(lh, ids_shape, y_shape, cond_names=(), ragged=True, row_round=512, pad_tx=16, pad_ty=32, ty_boundaries=None):
    """-> (key, padded T_x, padded T_y) of a batch: which captured graph can run it.  Padded T_x is the next multiple of pad_tx; padded
    T_y the next multiple of pad_ty, or — with ty_boundaries, the length boundaries of the bucketed sampler that builds the batches
    (data.DistributedBucketSampler) — the smallest boundary that holds the batch; the ragged row counts (text rows, squeezed mel
    rows, frame rows) are rounded up to row_round.  Coarser paddings mean fewer keys: SAMPLER_CAPTURE_CONFIG below."""
    from . import ops
    Tx = -(-int(ids_shape[-1]) // pad_tx) * pad_tx
    Ty = -(-int(y_shape[-1]) // pad_ty) * pad_ty
    if ty_boundaries:
        import bisect
        bs = sorted(-(-int(b) // 2) * 2 for b in ty_boundaries)
        i = bisect.bisect_left(bs, int(y_shape[-1]))
        if i < len(bs):
            Ty = -(-bs[i] // pad_ty) * pad_ty
    rows = (0, 0)
    if ragged:
        _, rx = ops.RowsCtx.row_starts(lh[0], Tx, row_round)
        _, ry = ops.RowsCtx.row_starts([int(v) // 2 for v in lh[1]], Ty // 2, row_round)
        _, rf = ops.RowsCtx.row_starts([int(v) // 2 * 2 for v in lh[1]], Ty // 2 * 2, row_round)
        rows = (rx, ry, rf)
    return rows + (tuple(ids_shape[:-1]) + (Tx,), tuple(y_shape[:-1]) + (Ty,), tuple(sorted(cond_names))), Tx, Ty


def sampler_capture_config(boundaries):
    """Trainer keyword arguments for batches built by a length-bucketed sampler with these boundaries (the reference trains with
    [32, 300, ..., 1000], train_ms_emo_lang_pitch.py:101-109): T_y padded to the bucket's boundary, T_x to 64, rows to 1024 (~3 % more
    masked rows than 512), a key captured the second time it is seen, 16 resident graphs.  On an LJSpeech-shaped length distribution
    that is 27 distinct keys, 29 captures in three epochs (1 233 steps) and a 97 % replay rate; the round-2 defaults (16 / 32 / 512,
    capture at first sight, 8 graphs) re-captured on 59 % of the steps (tests/test_capture_policy.py)."""
    return dict(ty_boundaries=list(boundaries), pad_tx=64, row_round=1024, capture_after=2, max_graphs=16)


class CapturePolicy:
    """Which graph keys are worth a capture: a key is admitted the `capture_after`-th time it is seen while not captured (host-side
    bookkeeping only; `seen` is bounded)."""

    def __init__(self, capture_after=2, max_tracked=4096):
        self.capture_after, self.max_tracked = max(1, int(capture_after)), int(max_tracked)
        self.seen = {}

    def admit(self, key):
        n = self.seen.get(key, 0) + 1
        if n < self.capture_after:
            if len(self.seen) >= self.max_tracked:
                self.seen.clear()
            self.seen[key] = n
            return False
        self.seen.pop(key, None)
        return True


class Trainer:
    """zero_grad -> forward -> loss -> backward -> all-reduce -> grad-norm -> AdamW step.

    graph=True captures the step (forward, MAS, backward, optimizer: several hundred kernel launches) into HIP graphs
    and replays them; the batch then lives in static buffers (`step` copies into them) and dropout masks still change
    every replay because every dropout kernel mixes the device-resident seed word (ops.seed_word) that the graph itself
    bumps.  One process alone: ONE graph.  Data-parallel (world > 1): the backward is phased (everything but the text
    encoder, then the text encoder) and the step is three graphs — forward + first backward | encoder backward |
    optimizer — with the RCCL all-reduces of the flat gradient buffer launched between them: the decoder's slice (~90 %
    of the bytes) is on the wire while the encoder's backward runs (collectives stay outside the captured regions).

    A first-seen graph key is side-effect free: the eager warm-up steps that precede a capture run without
    collectives, and parameters, Adam moments, the step word and the dropout seed word are restored afterwards — every
    `step()` applies exactly ONE optimizer update and issues exactly one step's collectives, whether it captured or
    replayed, so ranks that meet new keys on different steps stay in lock-step.  Batches are padded (zeros, masked by the
    lengths) to multiples of (pad_tx, pad_ty) so that the number of distinct keys stays small, and at most `max_graphs`
    captured keys are kept (least recently used goes first; ~2 GiB of graph memory each at cfg 2).

    Rows layout: ragged (model.rows_cfg.ragged; GT_RAGGED=0 or ragged=False turns it off) — every utterance owns exactly
    its own frames, so the ~30 % of padded frames of an LJSpeech-shaped batch cost nothing.  The row count is rounded
    (128 eager, 512 under graphs) and one graph is captured per distinct (text rows, mel rows, padded shapes) key; a
    replay only refreshes the per-utterance row offsets on the device.  `step(..., lengths_host=(x_lengths, y_lengths))`
    takes the lengths as Python ints (the data loader has them); without it they are read back from the device (one
    sync per step)."""

    WARMUPS = 2

    def __init__(self, model, lr=2e-4, betas=(0.9, 0.98), eps=1e-9, world=1, graph=False, total_steps=None,
                 split_graph=None, ragged=None, max_graphs=8, pad_tx=16, pad_ty=32, kernel_stamps=False, grad_wire="fp32",
                 force_collectives=False, capture_after=2, ty_boundaries=None, row_round=None, early_decoder_adam=True, pack_in_tail=True, split_roots=True):
        """total_steps: length of the OneCycleLR schedule the reference runs (train_ms_emo_lang_pitch.py:161);
        None keeps lr / betas constant.  split_graph=True selects the phased form (the decoder's gradient slice on the wire while the
        encoder's backward runs: three graphs); the default at any world size is ONE backward — the encoder's backward beside the
        decoder's, which is worth 1.16 ms of a 4.84 ms step (bench.py --split-graph 1 at N = 1: 6.0 ms) — and then the whole flat
        buffer on the wire (114 MB fp32: ~0.3-0.7 ms exposed on an 8-GPU xGMI node) between the two graphs."""
        from collections import OrderedDict
        self.model = model
        self.world = world
        # early_decoder_adam: without collectives, the optimizer's pass over the decoder's conv parameters starts right behind the
        # decoder's batched weight gradients (_early_decoder_update) instead of after the whole backward; same numbers either way
        self.early_decoder_adam = bool(early_decoder_adam)
        self.pack_in_tail = bool(pack_in_tail)
        # cfg-5-like models (stochastic predictors on the encoder's stream): the single-backward step does not join that branch at the
        # end of the forward; the backward gets one root per stream (_loss_roots)
        self.split_roots = bool(split_roots)   # ... followed by the next step's packing of the decoder's weights (see _early_decoder_update)
        self._dec_fresh_version = None        # flat_p._version at which the decoder's packed weight images were last made at a step's end
        self._head_rest = self._tail_packed = False
        self.graph_mode = bool(graph)
        self.split = bool(split_graph)
        accum = []
        for mod in model.modules():
            kind = type(mod).__name__
            if kind in ("LayerNorm", "ActNorm", "InvConvNear", "Embedding"):
                accum += list(mod.parameters(recurse=False))
            elif kind == "MultiHeadAttention":
                accum += [mod.emb_rel_k, mod.emb_rel_v]
        # flat-buffer order: everything but the decoder first, the decoder's parameters last (whatever the registration order
        # of the model's sub-modules; checkpoint.py maps the flat layout back to model.parameters() order by identity)
        named = list(model.named_parameters())
        # ... and inside the decoder the parameters its batched weight-gradient flush does NOT write (the conditioning layers, whose
        # effective weights are formed by host-side torch ops: their gradients arrive through autograd at the end of the backward)
        # before the ones it does: the tail of the buffer is then final the moment that flush is (_early_decoder_update)
        late = lambda n: ".cond_layer." in n or "cond_layer1" in n                 # noqa: E731
        plist = [p for n, p in named if not n.startswith("decoder.")] + [p for n, p in named if n.startswith("decoder.") and late(n)] + \
                [p for n, p in named if n.startswith("decoder.") and not late(n)]
        self.buckets = GradBuckets(plist, world, accum=accum, wire=grad_wire,      # grad_wire="bf16": GradBuckets.__init__
                                   force_collectives=force_collectives)
        # phased backward: the decoder's parameters (the tail of the flat buffer, ~90 % of the bytes) are final after the
        # first backward call and travel over xGMI while the text encoder's backward runs
        name_of = {id(p): n for n, p in model.named_parameters()}
        names = [name_of[id(p)] for p in self.buckets.params]                 # flat-buffer order: accumulated ones first
        self.dec0 = next((i for i, n in enumerate(names) if i >= self.buckets.n_accum and n.startswith("decoder.")), len(names))
        assert all(n.startswith("decoder.") for n in names[self.dec0:]), "decoder parameters must be the tail of the model"
        self.dec0_off = self.buckets.offsets[self.dec0] if self.dec0 < len(names) else self.buckets.total
        # the part of that tail the decoder's weight-gradient flush completes (early optimizer pass, packing at the step's end)
        self.dec_cov = next((i for i, n in enumerate(names) if i >= self.dec0 and not late(n)), len(names))
        self.dec_cov_off = self.buckets.offsets[self.dec_cov] if self.dec_cov < len(names) else self.buckets.total
        self.opt = FlatAdamW(self.buckets, lr, betas, eps)
        self.max_lr, self.total_steps, self.n_steps = lr, total_steps, 0
        self.grad_norm = None
        self._grad_norm_buf = None
        self._captured = OrderedDict()            # key -> (graphs, static inputs, outputs, row contexts); LRU, <= max_graphs
        self.max_graphs = int(max_graphs)
        # Capture policy.  A first-seen graph key costs two eager warm-up passes + the capture (~3 steps of work) and ~2 GiB of graph
        # memory, and a length-bucketed sampler produces many more (padded T_x, padded T_y, rounded row counts) combinations than
        # max_graphs slots: a key is captured only once it has been seen `capture_after` times (until then its steps run eagerly,
        # with exactly one step's collectives), so that one-off shapes do not evict the graphs of the frequent ones.
        # `capture_stats()` reports the hit rate.
        self.policy = CapturePolicy(capture_after)
        self.n_replays = self.n_eager = 0
        self.pad_tx, self.pad_ty = int(pad_tx), int(pad_ty)
        assert self.pad_ty % 2 == 0
        self.n_captures = 0
        self.cfg = model.rows_cfg                 # this model's rows-layout state (ops.RowsConfig): nothing process-global
        self.cfg.ragged = (os.environ.get("GT_RAGGED", "1") != "0") if ragged is None else bool(ragged)
        # ragged row count granularity (one graph per rounded size)
        self.cfg.row_round = int(row_round) if row_round else (512 if self.graph_mode else 128)
        self.ty_boundaries = list(ty_boundaries) if ty_boundaries else None
        # bench.py: device-side begin / end stamps of every fused WaveNet-layer forward launch, valid inside replayed graphs
        self.stamps = None
        if kernel_stamps:
            from . import ops
            dec = model.decoder
            per_step = dec.n_blocks * dec.n_layers * (3 if hasattr(dec.flows[2], "wn_pitch") else 1)
            self.stamps = self.cfg.stamps = ops.KernelStamps(next(model.parameters()).device, per_step)

    @property
    def adam_steps(self):
        """Adam's step word on the device (equals n_steps: one update per step(); reads back, test / logging use)."""
        return int(self.opt.hyper[5].item())

    def _loss(self, outs):
        """The reference's training loss (train_ms_emo_lang_pitch.py:295-306): mle + sum(l_length) [+ 0.5 l_pitch + 0.5 l_energy]."""
        m = self.model
        (z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, l_pitch, l_energy), _, _ = outs
        l_mle = models.mle_loss(z, z_m, None if m.mean_only else z_logs, logdet, z_mask)
        loss = l_mle + torch.sum(l_length)
        if l_pitch is not None:
            loss = loss + 0.5 * l_pitch
        if l_energy is not None:
            loss = loss + 0.5 * l_energy
        return loss, l_mle

    def _loss_roots(self, outs):
        """_loss() for a forward that did not join the predictors' branch (ops.RowsConfig.join_predictors = False): one backward root
        per stream — the likelihood term (+ the energy term when its chain ran on this stream) here, the duration and pitch terms on
        the encoder's stream, where they were produced — so that nothing on this stream waits for the predictors' forward before the
        decoder's backward starts.  -> ([roots], l_mle); their sum is _loss()'s loss."""
        m = self.model
        ls = getattr(m, "_loss_streams", None)
        if ls is None:
            loss, l_mle = self._loss(outs)
            return [loss], l_mle
        (z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, l_pitch, l_energy), _, _ = outs
        l_mle = models.mle_loss(z, z_m, None if m.mean_only else z_logs, logdet, z_mask)
        here = l_mle
        with torch.cuda.stream(ls["stream"]):
            there = torch.sum(l_length)
            if l_pitch is not None:
                there = there + 0.5 * l_pitch
            if l_energy is not None and not ls["energy_on_caller"]:
                there = there + 0.5 * l_energy
        if l_energy is not None and ls["energy_on_caller"]:
            here = here + 0.5 * l_energy
        return [here, there], l_mle

    def _begin(self, device):
        from . import ops
        if self.stamps is not None:
            self.stamps.begin_step()
        if ops.MARKS is not None:
            ops.MARKS.names = []
            ops.mark("step begin")
        # The step's head: three accumulator fills, the gathers of the derived biases and the weight-packing launch, all before
        # either branch can start.  Only the packing is long (~100 us): the rest runs beside it on the encoder's stream.
        from . import text_models
        side = None
        if device.type == "cuda" and self.cfg.encoder_stream and hasattr(self.model, "prepare"):
            side = text_models._encoder_stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            # ONE launch (gt_step_zero): the arena of the step's small zeroed accumulators, the region of the partly written per-layer
            # buffers (attention), the atomically accumulated parameter gradients' slice of the flat buffer, and the dropout seed's bump
            ops.step_head(device, extra=self.buckets.accum_region())
        for p in self.buckets.params:
            p.grad = None
        self._head_rest = False
        if side is None:
            self._dec_fresh_version = None
        if side is not None:
            # the decoder's weights were packed at the end of the previous step (_early_decoder_update) unless anything touched the
            # parameters since (torch bumps the flat buffer's version counter on every in-place op on a view of it; the
            # optimizer and packing kernels go through raw pointers and do not)
            self._head_rest = self._dec_fresh_version == self._param_version()
            # (no join here: FlowGenerator.forward makes its stream wait for `side` behind its conditioning front end — cfg 5's
            #  speaker / emotion / language embeddings are ~40 small launches that need no packed weight)
            self.model.prepare(side=side, part="rest" if self._head_rest else "all", join=False)
            self.model._prepared_by_trainer = True
            self.model._prepare_side = side
            self._dec_fresh_version = None
        ops.mark("accumulators zeroed")

    def _param_version(self):
        """Changes whenever torch wrote into the decoder's parameters (load_state_dict, the capture pass's restore, an in-place op on a
        parameter): in-place ops bump a tensor's version counter, the optimizer and packing kernels go through raw pointers and do
        not.  (A write through `p.data` is invisible to it: call invalidate_packed() after one.)"""
        return self.opt.flat_p._version + sum(p._version for p in self.buckets.params[self.dec_cov:])

    def invalidate_packed(self):
        """The decoder's packed weight images no longer match its parameters: the next step packs everything at its head."""
        self._dec_fresh_version = None

    def _early_decoder_update(self, params_done):
        """Called by the decoder's backward right behind its batched weight gradients (wgrad.WgradQueue.flush, site = the decoder): if
        they covered every parameter of the flat buffer's tail — the decoder's conv weights, gains and biases — the optimizer's pass
        over that tail starts now, on the decoder's stream, while the text encoder's branch is still in its backward."""
        done = {id(p) for p in params_done}
        if all(id(p) in done for p in self.buckets.params[self.dec_cov:]) and self.dec_cov_off < self.buckets.total:
            self.opt.step_early(self.dec_cov_off, self.buckets.total)
            if self.pack_in_tail and self.cfg.encoder_stream and hasattr(self.model, "prepare"):
                # ... and the NEXT step's packing of these weights (90 % of that launch) follows at once, still beside the encoder's
                # branch: the next step's head then packs the rest only.  (The decoder's other parameters — ActNorm, InvConvNear —
                # are not part of any packed image.)
                self.model.prepare(part="decoder")
                self._tail_packed = True
                self._dec_fresh_version = self._param_version()

    def _fwd_bwd(self, ids, t_x, y, t_y, lengths_host=None, cond=None, early_update=False):
        """forward + the whole backward; early_update (a full step without collectives, _step_impl): the optimizer's pass over the
        decoder's parameters is launched from inside the backward (_early_decoder_update) and opt.step() must follow."""
        self._begin(ids.device)
        from . import ops
        rows_cfg = getattr(self.model, "rows_cfg", None)
        if rows_cfg is not None:
            rows_cfg.join_predictors = not self.split_roots
        try:
            roots, l_mle = self._loss_roots(self.model(ids, t_x, y, t_y, lengths_host=lengths_host, **(cond or {})))
        finally:
            if rows_cfg is not None:
                rows_cfg.join_predictors = True
        ops.mark("loss")
        dec = getattr(self.model, "decoder", None)
        early = early_update and self.early_decoder_adam and dec is not None and not self.buckets.collect and ids.is_cuda
        if early:                                    # (with collectives the gradients are not final until the all-reduce)
            object.__setattr__(dec, "_gt_after_flush", self._early_decoder_update)
        try:
            if len(roots) == 1:
                roots[0].backward()
            else:
                torch.autograd.backward(roots)
        finally:
            if early:
                object.__setattr__(dec, "_gt_after_flush", None)
        if len(roots) > 1:                           # (the engine has joined the streams its leaves were accumulated on; make it explicit)
            torch.cuda.current_stream().wait_stream(self.model._loss_streams["stream"])
            loss = roots[0].detach() + roots[1].detach()
        else:
            loss = roots[0]
        ops.mark("backward joined")
        self.buckets.gather()
        ops.mark("gradients gathered")
        return loss.detach(), l_mle.detach()

    def _optim(self, device):
        from . import ops
        # reference commons.clip_grad_value_(params, None): total grad norm, no clipping — the sum of squares
        # falls out of the optimizer's own pass over the gradients (no ~1.8k .item() syncs)
        # into ONE persistent tensor: a replayed graph runs no Python, so a tensor created here would be the last CAPTURED
        # graph's, stale whenever another row bucket's graph is the one replaying
        if self._grad_norm_buf is None:
            self._grad_norm_buf = torch.zeros(1, dtype=torch.float32, device=device)
        self.grad_norm = torch.sqrt(self.opt.step(), out=self._grad_norm_buf)
        ops.mark("optimizer done")
        if self.stamps is not None:
            self.stamps.end_step()
        ops.arena_end(device)

    def _phase1(self, ids, t_x, y, t_y, lengths_host=None, cond=None):
        """forward + the backward of everything but the text encoder; the decoder's gradients are then in the flat buffer."""
        self._begin(ids.device)
        loss, l_mle = self._loss(self.model(ids, t_x, y, t_y, lengths_host=lengths_host, defer_encoder_backward=True,
                                            **(cond or {})))
        loss.backward()
        self.buckets.gather(self.dec0, None)
        return loss.detach(), l_mle.detach()

    def _phase2(self):
        self.model.backward_encoder()
        self.buckets.gather(0, self.dec0)

    def _step_impl(self, ids, t_x, y, t_y, lengths_host=None, cond=None, collectives=True):
        if not self.split:
            out = self._fwd_bwd(ids, t_x, y, t_y, lengths_host, cond, early_update=True)
            if collectives:
                self.buckets.allreduce()
        else:
            out = self._phase1(ids, t_x, y, t_y, lengths_host, cond)
            if collectives:
                self.buckets.allreduce(self.dec0_off, None, wait=False)      # on the wire while the encoder's backward runs
            self._phase2()
            if collectives:
                self.buckets.allreduce(0, self.dec0_off, wait=True)
        self._optim(ids.device)
        return out

    # ---- capture ---------------------------------------------------------------------------------------------------
    def _snapshot(self, device):
        from . import ops
        o = self.opt
        return [(t, t.clone()) for t in (o.flat_p, o.m, o.v, o.hyper, ops.seed_word(device))]

    @staticmethod
    def _restore(snap):
        for t, c in snap:
            t.copy_(c)

    def _capture(self, ids, t_x, y, t_y, lh, cond=None):
        from . import ops
        static = [t.clone() for t in (ids, t_x, y, t_y)] + [{k: v.clone() for k, v in (cond or {}).items()}]
        ctxs = {}
        if self.cfg.ragged:                          # row contexts live outside the graph; replays refresh them in place
            ctxs["x"] = ops.RowsCtx(static[1].to(torch.int32), ids.shape[1], lengths_host=lh[0], round_to=self.cfg.row_round)
            ctxs["y"] = ops.RowsCtx((static[3] // 2).to(torch.int32), y.shape[2] // 2, lengths_host=[int(v) // 2 for v in lh[1]],
                                    round_to=self.cfg.row_round)
            if getattr(self.model, "use_spp", False) or getattr(self.model, "use_sep", False):     # frame-rate rows of the pitch / energy predictors
                ctxs["f"] = ops.RowsCtx((static[3] // 2 * 2).to(torch.int32), y.shape[2] // 2 * 2,
                                        lengths_host=[int(v) // 2 * 2 for v in lh[1]], round_to=self.cfg.row_round)
        self.cfg.prebuilt.update(ctxs)
        try:
            return self._capture_with(static, lh, ctxs)
        finally:
            self.cfg.prebuilt.clear()

    def _capture_with(self, static, lh, ctxs):
        ids = static[0]
        cond = static[4]
        static = static[:4]
        cur = torch.cuda.current_stream()
        if getattr(self, "_cap_stream", None) is None:   # ONE capture stream per trainer: the parameters' AccumulateGrad
            self._cap_stream = torch.cuda.Stream()       # nodes stay bound to the stream of the first warm-up
        side = self._cap_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            # warm-up (one-time attribute calls, scratch growth, pinned staging pools) WITHOUT side effects: no
            # collectives, and everything the optimizer / dropout state machine touched is put back afterwards
            snap = self._snapshot(ids.device)
            try:
                for _ in range(self.WARMUPS):
                    self._step_impl(*static, lengths_host=lh, cond=cond, collectives=False)
            finally:
                self._restore(snap)
            if self._tail_packed and hasattr(self.model, "prepare"):
                # the warm-up steps packed the decoder's weights at their ends — from parameters the restore has just undone: pack them
                # again from the restored ones, so that the captured step (whose head then packs the rest only) finds them fresh
                self.model.prepare(part="decoder")
                self._dec_fresh_version = self._param_version()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._tail_packed = False
        # capture on the stream the warm-up ran on: autograd's AccumulateGrad nodes remember the stream they were
        # created on, and work they launched on another stream would stay outside the captured graph
        from . import wgrad
        tables = wgrad.table_arena_begin(ids.device) # the captured weight-gradient kernels' job tables: outside the graph's pool
        g1 = torch.cuda.CUDAGraph()
        if not self.split and not self.buckets.collect:
            with torch.cuda.graph(g1, stream=side, capture_error_mode=CAPTURE_MODE):
                out = self._step_impl(*static, lengths_host=lh, cond=cond, collectives=False)
            graphs = (g1,)
        elif not self.split:                         # one backward, then the whole buffer on the wire, then the optimizer
            with torch.cuda.graph(g1, stream=side, capture_error_mode=CAPTURE_MODE):
                out = self._fwd_bwd(*static, lengths_host=lh, cond=cond)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
                self._optim(ids.device)
            graphs = (g1, g2)
        else:
            with torch.cuda.graph(g1, stream=side, capture_error_mode=CAPTURE_MODE):
                out = self._phase1(*static, lengths_host=lh, cond=cond)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
                self._phase2()
            g3 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g3, stream=side, pool=g1.pool(), capture_error_mode=CAPTURE_MODE):
                self._optim(ids.device)
            graphs = (g1, g2, g3)
        self.n_captures += 1
        wgrad.sync_uploads(ids.device)               # fill the tables the captured kernels read (once, not per replay)
        wgrad.table_arena_end(ids.device)
        ctxs = dict(ctxs or {}, _keep=tables)        # wgrad.CaptureKeep: lives (and is released) with this key's graphs
        # how the captured step packs the decoder's weights: its head packs everything (False) or relies on the previous step's
        # end having packed the decoder's (True); its own end packs them for the next step or not
        ctxs["_head_rest"], ctxs["_tail_packed"] = self._head_rest, self._tail_packed
        return graphs, static + [cond], out, ctxs

    def _rows_key(self, Tx, Ty, lh):
        """the rounded ragged row counts (text, squeezed mel, frame rows) of a batch padded to (Tx, Ty): the head of its graph key"""
        return graph_key(lh, (len(lh[0]), Tx), (len(lh[1]), 1, Ty), (), self.cfg.ragged, self.cfg.row_round, 1, 2)[0][:3]

    @staticmethod
    def _pad_time(t, T):
        """zero-pad the last dim of t to T (the lengths mask the padding everywhere downstream)"""
        if t.shape[-1] == T:
            return t
        out = t.new_zeros(t.shape[:-1] + (T,))
        out[..., :t.shape[-1]] = t
        return out

    def precapture(self, batches):
        """Capture the graphs of these batches up front ([(ids, t_x, y, t_y, lengths_host, cond kwargs)]) without taking
        a training step: parameters / optimizer state are exactly what they were afterwards."""
        if not self.graph_mode:
            return
        for ids, t_x, y, t_y, lh, cond in batches:
            self._graph_for(ids, t_x, y, t_y, lh, dict(cond or {}), force=True)

    def capture_stats(self):
        """{captures, replays, eager steps, hit rate = replays / steps taken in graph mode, graphs resident}"""
        n = self.n_replays + self.n_eager
        return {"captures": self.n_captures, "replays": self.n_replays, "eager": self.n_eager,
                "hit_rate": (self.n_replays / n) if n else None, "resident": len(self._captured)}

    def _evict(self):
        """least recently used key goes: its graphs free their pool, its CaptureKeep returns the pinned staging buffers to the pool
        and drops the tables / scratch buffers only those graphs read"""
        _, (graphs, static, out, ctxs) = self._captured.popitem(last=False)
        keep = ctxs.pop("_keep", None) if isinstance(ctxs, dict) else None
        del graphs, static, out, ctxs
        if keep is not None:
            keep.release()

    def _graph_for(self, ids, t_x, y, t_y, lh, cond, force=False):
        """-> (captured entry or None if this step runs eagerly (a cold key, or capture is impossible), padded inputs)"""
        key, Tx, Ty = graph_key(lh, ids.shape, y.shape, cond, self.cfg.ragged, self.cfg.row_round, self.pad_tx, self.pad_ty,
                                self.ty_boundaries)
        cap = self._captured.get(key)
        if cap is None and not force and not self.policy.admit(key):
            return None, (ids, y, cond)              # a cold key: this step runs eagerly
        if cap is None:
            # a new key: the capture clones PADDED inputs (a known key's batch goes straight into the padded static buffers)
            ids, y = self._pad_time(ids, Tx), self._pad_time(y, Ty)
            cond = {k: (self._pad_time(v, Ty) if k in ("pitch", "energy") else v) for k, v in cond.items()}
            try:
                cap = self._capture(ids, t_x, y, t_y, lh, cond)
            except Exception as e:                   # e.g. an allocation the capture refuses: keep training, eagerly
                import warnings
                warnings.warn(f"HIP graph capture of the training step failed ({e!r}); continuing with eager launches")
                self.graph_mode = False
                # auxiliary streams that had joined the aborted capture may be left in capture state: start over with fresh ones
                from . import text_models, wgrad
                text_models._ENC_STREAMS.clear(); wgrad._SIDE.clear(); wgrad._PENDING.clear()
                self.cfg.prebuilt.clear()
                return None, (ids, y, cond)
            self._captured[key] = cap
            while len(self._captured) > self.max_graphs:
                self._evict()
        else:
            self._captured.move_to_end(key)
        return cap, (ids, y, cond)

    def _ddi_pass(self, ids, t_x, y, t_y, lh, cond):
        """ActNorm's data-dependent init (ActNorm.set_ddi(True); modules.py:604-619, the reference's init.py:17-23 runs it as ONE
        forward pass and saves the result): if any ActNorm is still uninitialised, this batch initialises them all, block after
        block, in a forward-only pass — no gradient, no optimizer update, outside any capture and outside the capture's
        snapshot / restore (inside it, the restored parameter buffer silently undid the initialisation while `initialized`
        stayed True).  The regular step on the same batch follows."""
        dec = getattr(self.model, "decoder", None)
        if dec is None or all(dec.flows[3 * b].initialized for b in range(dec.n_blocks)):
            return
        assert not torch.cuda.is_current_stream_capturing()
        was = self.model.training
        with torch.no_grad():
            self.model(ids, t_x, y, t_y, lengths_host=lh, **(cond or {}))
        self.model.train(was)
        assert all(dec.flows[3 * b].initialized for b in range(dec.n_blocks))

    def step(self, ids, t_x, y, t_y, lengths_host=None, g=None, pitch=None, energy=None, l=None, emo=None, emo_cartesian=None):
        """One optimizer step.  g: speaker input ([b, gin_channels, 1] at the encoder boundary, or the raw [b, 512]
        embedding when the model owns emb_g); pitch / energy [b, 1, t_y]: raw contours of cfg 5 (FlowGenerator.forward
        normalises them, models.py:1054-1071); l [b]: language ids; emo [b] / emo_cartesian [b, 3]: cfg 5's emotion inputs."""
        cond = {k: v for k, v in (("g", g), ("pitch", pitch), ("energy", energy), ("l", l), ("emo", emo),
                                  ("emo_cartesian", emo_cartesian)) if v is not None}
        if self.total_steps:
            self.opt.set_schedule(*one_cycle(self.n_steps, self.total_steps, self.max_lr))
        self.n_steps += 1
        lh = lengths_host
        if self.cfg.ragged and lh is None:
            lh = (t_x.tolist(), t_y.tolist())        # device sync: pass lengths_host to avoid it
        self._ddi_pass(ids, t_x, y, t_y, lh, cond)
        if not self.graph_mode:
            return self._step_impl(ids, t_x, y, t_y, lh, cond=cond)
        cap, (ids_p, y_p, cond_p) = self._graph_for(ids, t_x, y, t_y, lh, cond)
        if cap is None:
            self.n_eager += 1
            if not self.graph_mode or not ids.is_cuda:           # (a failed capture switched graph mode off)
                return self._step_impl(ids, t_x, y, t_y, lh, cond=cond)
            # A cold key's eager step runs on the CAPTURE stream: autograd's AccumulateGrad nodes remember the stream of the first
            # backward they saw, and a later capture whose backward has to synchronise with a stream outside the capture (the
            # default stream an eager step would have bound them to) is an invalid capture — hipStreamEndCapture answered that with
            # a segmentation fault (cfg 5, round 3; the round-2 "four capture streams" crash was the same class: a stream the
            # capture had to wait for without it ever having joined the capture).
            cur = torch.cuda.current_stream()
            if getattr(self, "_cap_stream", None) is None:
                self._cap_stream = torch.cuda.Stream()
            side = self._cap_stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out = self._step_impl(ids, t_x, y, t_y, lh, cond=cond)
            cur.wait_stream(side)
            for t_ in (ids, t_x, y, t_y) + tuple((cond or {}).values()):
                t_.record_stream(side)
            return out
        self.n_replays += 1
        graphs, static, out, ctxs = cap
        pairs = list(zip(static[:4], (ids_p, t_x, y_p, t_y))) + [(static[4][k], v) for k, v in cond_p.items()]
        ctx_lens = []
        if ctxs and "x" in ctxs:                        # per-utterance row offsets / masks of THIS batch (same rounded size)
            ctx_lens = [(ctxs["x"], lh[0]), (ctxs["y"], [int(v) // 2 for v in lh[1]])]
            if "f" in ctxs:
                ctx_lens.append((ctxs["f"], [int(v) // 2 * 2 for v in lh[1]]))
        _upload_step_inputs(pairs, ctx_lens)
        if ctxs.get("_head_rest") and self._dec_fresh_version != self._param_version():
            self.model.prepare(part="decoder")       # something touched the parameters since the last step's end (or that step did not pack)
        graphs[0].replay()                           # collectives sit BETWEEN the graphs, never inside one
        if len(graphs) == 2:
            self.buckets.allreduce()
            graphs[1].replay()
        elif len(graphs) == 3:
            self.buckets.allreduce(self.dec0_off, None, wait=False)      # decoder slice: overlaps the encoder's backward
            graphs[1].replay()
            self.buckets.allreduce(0, self.dec0_off, wait=True)
            graphs[2].replay()
        self._dec_fresh_version = self._param_version() if ctxs.get("_tail_packed") else None
        return out
