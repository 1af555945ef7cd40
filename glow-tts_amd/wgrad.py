"""Deferred, batched parameter gradients of the rows-layout convolutions.

The reference gets every conv's weight/bias gradient from autograd, one ATen call per module as the
backward walks the graph (modules.py:127-171, attentions.py:103-259).  On one MI355X a single conv's
wgrad is a small GEMM (a few GFLOP) that cannot fill 256 CUs, and its weight-norm backward is one more
tiny launch.  HBM is plentiful (288 GB), so the backward keeps every layer's input rows and output-gradient
rows alive, runs only the data-gradient chain, and at the end `WgradQueue.flush()` computes ALL weight
gradients with one launch per tap count (`gt_conv_wgrad_batched`) and ALL weight-norm Jacobians / bias
gradients with one more (`gt_weightnorm_bwd_batched`).
"""
import os

import numpy as np
import torch

from . import _lib

JOB = np.dtype([("X", "u8"), ("dY", "u8"), ("part", "u8"), ("part_bias", "u8"), ("ldx", "i4"), ("ldy", "i4"), ("R", "i4"),
                ("Cin", "i4"), ("Cout", "i4"), ("co_begin", "i4"), ("co_count", "i4"), ("slab_rows", "i4")])
TILE = np.dtype([("job", "i4"), ("co0", "i4"), ("ci0", "i4"), ("slab", "i4")])
WNB = np.dtype([("part", "u8"), ("part_bias", "u8"), ("v", "u8"), ("g", "u8"), ("inv_norm", "u8"), ("dv", "u8"), ("dg", "u8"),
                ("dbias", "u8"), ("S", "i4"), ("Cout", "i4"), ("Cin", "i4"), ("taps", "i4"), ("row_start", "i4"),
                ("accumulate", "i4"), ("pad0_", "i4"), ("pad1_", "i4")])
assert JOB.itemsize == 64 and TILE.itemsize == 16 and WNB.itemsize == 96

# Row slabs per job.  A launch's workgroups (tiles x slabs) run two per CU — SLOTS at a time —, each walks its slab's rows at about
# ROW_US per row whatever the tap count (operand latency, not MFMA, sets it; measured 0.033 us with both slots of a CU busy), and every
# slab costs one more copy of dW written here and re-read by the weight-norm backward (PART_BPS).  choose_slabs() takes the cheapest
# count under that model; it reproduces the measured optima of round 3 (`profiles/r03_wgrad_variants.txt`): 1 slab for the decoder's 432
# k = 5 tiles (2 slabs: same kernel time, 35 us more partial traffic), 2 for its 180 k = 1 tiles (1: 186 us, 2: 147 us, 3: 160 us).
SLOTS = 512
ROW_US = float(os.environ.get("GT_WGRAD_ROW_US", "0.033"))         # dev knobs
WG_US = 3.0                            # prologue + partial stores of one workgroup
PART_BPS = 2.0e6                       # bytes of slab partials per microsecond, write + re-read
MAX_SLABS = 32
XCDS = 8
CI_TILE = {5: 64, 3: 64, 1: 192}      # input channels per workgroup tile (conv_wgrad.hip WgSel; checked against the library at flush time)
FORCE_SLABS = int(os.environ.get("GT_WGRAD_SLABS", "0"))           # dev knob: the same slab count for every job

# ASYNC (process-level dev knob GT_WGRAD_ASYNC=1, read once at import; off by default): flush() launches on a side stream, so the batched weight-gradient kernels of the
# decoder overlap with whatever the backward does next (the rest of the data-gradient chain, the text encoder's
# backward: mostly small kernels that leave CUs idle); join() makes the current stream wait for them and must run
# before anything reads the gradients (train.GradBuckets.gather does).
ASYNC = os.environ.get("GT_WGRAD_ASYNC", "0") != "0"   # measured: a third concurrent stream costs more than it hides
_SIDE = {}
_PENDING = set()
_ACTIVE = []            # stack of open queues
_KEEP = []              # tables referenced by captured graphs whose queue had no owning module (site)
_SCRATCH = {}


_POOL = []              # pinned staging buffers set aside (outside capture) for use inside a graph capture
_POOL_N, _POOL_BYTES = 64, 1 << 18


def _fill_pool():
    while len(_POOL) < _POOL_N:
        _POOL.append(torch.empty(_POOL_BYTES, dtype=torch.uint8).pin_memory())


_PENDING_UPLOADS = []        # (device table, pinned host copy) of the flushes recorded by the graph capture in progress
_TABLE_ARENA = {}            # device -> {"buf", "off"}: where a capture's tables live (see _flush)
_CAPTURE_KEEP = None         # while a trainer captures a step: what that capture's graphs read (pinned staging buffers, tables, scratch)


class CaptureKeep(list):
    """Everything ONE captured step reads besides its own memory pool: it lives in the trainer's entry for that graph key and goes
    with it — evicting the key (train.Trainer's LRU) returns the pinned staging buffers to the pool and drops the last reference
    to tables and outgrown scratch buffers.  (Hung on the module or on a module-level list, as before, every capture leaked its
    8 MB table arena and its pinned buffers for the life of the model.)"""

    def release(self):
        for obj in self:
            if isinstance(obj, torch.Tensor) and not obj.is_cuda and obj.numel() == _POOL_BYTES and obj.is_pinned() and len(_POOL) < 4 * _POOL_N:
                _POOL.append(obj)
        self.clear()


def table_arena_begin(dev, nbytes=8 << 20):
    """Before a graph capture (outside it): a buffer for the capture's weight-gradient tables.  Returns the CaptureKeep of this
    capture (the arena is its first element): the caller keeps it exactly as long as the graph and calls .release() on eviction."""
    global _CAPTURE_KEEP
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _TABLE_ARENA[str(dev)] = {"buf": buf, "off": 0}
    _PENDING_UPLOADS.clear()                          # nothing left over from a capture that failed half-way
    _CAPTURE_KEEP = CaptureKeep([buf])
    return _CAPTURE_KEEP


def table_arena_end(dev):
    global _CAPTURE_KEEP
    _TABLE_ARENA.pop(str(dev), None)
    _CAPTURE_KEEP = None


def capture_keep(obj):
    """Tie obj's lifetime to the graph capture in progress (a scratch buffer the captured kernels write into, say); False if no
    trainer-owned capture is open."""
    if _CAPTURE_KEEP is not None:
        if not any(o is obj for o in _CAPTURE_KEEP):
            _CAPTURE_KEEP.append(obj)
        return True
    return False



def sync_uploads(dev):
    """Right after a graph capture (outside it): fill the tables its weight-gradient kernels read; nothing else ever writes them."""
    if _PENDING_UPLOADS:
        assert not torch.cuda.is_current_stream_capturing()
        for dst, host in _PENDING_UPLOADS:
            dst.copy_(host, non_blocking=True)
        _PENDING_UPLOADS.clear()
        torch.cuda.current_stream(dev).synchronize()


def choose_slabs(base_tiles, R, part_bytes):
    """Slabs per job for a launch of base_tiles workgroup tiles (at one slab) over jobs of ~R rows whose weight gradients are
    part_bytes of fp32 in total: the smallest count within 5 % of the cheapest under the cost model above; slabs stay >= 128 rows."""
    if FORCE_SLABS:
        return FORCE_SLABS
    best, costs = None, []
    for S in range(1, max(1, min(MAX_SLABS, R // 128)) + 1):
        rounds = -(-base_tiles * S // SLOTS)
        rows = -(-(-(-R // S)) // 64) * 64
        cost = rounds * (rows * ROW_US + WG_US) + S * part_bytes / PART_BPS
        costs.append((cost, S))
        best = cost if best is None else min(best, cost)
    return min(S for cost, S in costs if cost <= 1.05 * best)


def _side_stream(dev):
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def join(dev):
    """Current stream waits for every asynchronously flushed queue of this device."""
    key = str(dev)
    if key in _PENDING:
        torch.cuda.current_stream(dev).wait_stream(_SIDE[key])
        _PENDING.discard(key)


def active():
    return _ACTIVE[-1] if _ACTIVE else None


def _scratch(dev, nbytes):
    """Partial-sum workspace of a flush, one per (device, stream the flush runs on): flushes on different streams (the
    decoder's and the text encoder's backward run concurrently, ops.RowsConfig.encoder_stream) must not share it."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() < nbytes:
        # A captured step has the address of the buffer it was captured with baked into its kernel arguments, and a later capture
        # (another row bucket, more slabs) may need a bigger one: the outgrown buffer must stay allocated for as long as any graph
        # can replay into it — freed, its pages go back to the allocator and the next replay of the older graph writes its partial
        # sums over whoever owns them then.  Every capture therefore holds a reference to the buffer it used (capture_keep below):
        # an outgrown buffer lives exactly as long as the graphs that replay into it, and no longer.
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        _SCRATCH[key] = buf
    if dev.type == "cuda" and torch.cuda.is_current_stream_capturing() and not capture_keep(buf):
        _KEEP.append(buf)                             # a capture nobody owns (no trainer): keep it for the life of the process
    return buf


def _grad_out(param):
    """Where a parameter's gradient is written: its slice of the trainer's flat gradient buffer when there is
    one (train.GradBuckets sets `_gt_flat_grad`; a fresh view, so autograd can adopt it without a copy)."""
    fg = getattr(param, "_gt_flat_grad", None)
    if fg is None:
        return torch.empty_like(param)
    buf, off = fg
    return buf[off:off + param.numel()].view_as(param)


class WgradQueue:
    """`with WgradQueue(dev) as q:` — conv_param_grads() inside only records (conv, x, dy) and hands back the
    (still unwritten) gradient tensors; leaving the block launches the batched kernels that fill them."""

    def __init__(self, device, site=None, accumulate=False, side_stream=False):
        self.dev = device
        self.site = site            # object that owns the cached tables (the runner's module)
        self.items = []             # (conv, R, parts[(x, dy, co_begin, co_count)], dv, dg, db)
        # accumulate: the gradients are ADDED to what their destinations hold; side_stream: flush on the wgrad side stream
        # even without ASYNC
        self.accumulate, self.force_side = accumulate, side_stream
        self.defer_to = None        # a list: leaving the block parks the queue there instead of flushing (the owner flushes later)
        self.ln_jobs = []           # (partials, dst_a, dst_b): parameter gradients left as per-workgroup partial rows, waiting for their one reduce launch
        self.params_done = []       # parameters whose gradients this queue's flush completes (handed to the site's _gt_after_flush)

    def __enter__(self):
        _ACTIVE.append(self)
        return self

    def __exit__(self, et, ev, tb):
        _ACTIVE.pop()
        if et is None:
            if self.defer_to is not None:
                self.defer_to.append(self)
            else:
                self.flush()
        return False

    def add(self, conv, R, parts, want_bias=True):
        v = conv.weight_v if conv.weight_norm else conv.weight
        dv = _grad_out(v)
        dg = _grad_out(conv.weight_g) if conv.weight_norm else None
        db = _grad_out(conv.bias) if (want_bias and conv.bias is not None) else None
        self.items.append((conv, R, parts, dv, dg, db))
        out = {v: dv}
        if dg is not None:
            out[conv.weight_g] = dg
        if db is not None:
            out[conv.bias] = db
        self.params_done.extend(out.keys())
        return out

    def add_ln(self, partials, dst_a, dst_b):
        """A backward kernel left its per-workgroup parameter-gradient sums in `partials` [rows, dst_a.numel() + dst_b.numel()]
        (gt_layernorm_bwd_partials: dgamma | dbeta; the DDSConv backward kernels: the same, and dw | db): they are added to dst_a / dst_b
        by ONE launch for all such buffers of the block (gt_param_partials_reduce) when the queue is flushed."""
        assert partials.shape[1] == dst_a.numel() + dst_b.numel()
        self.ln_jobs.append((partials, dst_a, dst_b))
        if len(self.ln_jobs) == _lib.PARTIALS_MAX:
            self._flush_ln()

    def _flush_ln(self):
        if not self.ln_jobs:
            return
        import ctypes
        args = _lib.PartialsArgs()
        for i, (part, da, db) in enumerate(self.ln_jobs):
            j = args.job[i]
            j.partials, j.dst_a, j.dst_b, j.n_rows, j.Ca, j.Cb = part.data_ptr(), da.data_ptr(), db.data_ptr(), part.shape[0], da.numel(), db.numel()
        args.n_jobs = len(self.ln_jobs)
        _lib.check(_lib.lib().gt_param_partials_reduce(ctypes.byref(args), _lib.current_stream(self.dev)), "gt_param_partials_reduce")
        self.ln_jobs = []

    def _plan(self):
        """-> (key, job rows, tile rows, wnb rows, counts, scratch bytes)"""
        jobs, wnbs, tiles = [], [], {5: [], 3: [], 1: []}
        off = 0
        row = 0
        max_n = 1
        # slabs: one count per launch (tap count) from choose_slabs() — a launch whose jobs are few and short (the text encoder's 3
        # pre-net convs, the duration predictor's 2) would otherwise run a dozen workgroups that each walk all the rows
        base, pbytes, rmax = {5: 0, 3: 0, 1: 0}, {5: 0, 3: 0, 1: 0}, {5: 1, 3: 1, 1: 1}
        for conv, R, parts, dv, dg, db in self.items:
            t = conv.pc.taps
            base[t] += sum(-(-cc // 128) * -(-conv.pc.Cin // CI_TILE[t]) for _, _, _, cc in parts)
            pbytes[t] += t * conv.pc.Cout * conv.pc.Cin * 4
            rmax[t] = max(rmax[t], R)
        slabs = {t: choose_slabs(base[t], rmax[t], pbytes[t]) if base[t] else 1 for t in (5, 3, 1)}
        for conv, R, parts, dv, dg, db in self.items:
            pc = conv.pc
            taps, Cin, Cout = pc.taps, pc.Cin, pc.Cout
            S = max(1, min(slabs[taps], R // 128 if not FORCE_SLABS else slabs[taps]))
            slab_rows = -(-(-(-R // S)) // 64) * 64
            S = -(-R // slab_rows)
            part_off, off = off, off + S * taps * Cout * Cin * 4
            pb_off, off = off, off + S * Cout * 4
            off = (off + 255) & ~255
            for x, dy, co_begin, co_count in parts:
                assert x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.stride(1) == 1 and dy.stride(1) == 1
                assert x.shape[0] >= R and dy.shape[0] >= R and x.shape[1] >= Cin and dy.shape[1] >= co_count
                jid = len(jobs)
                jobs.append((x.data_ptr(), dy.data_ptr(), part_off, pb_off if db is not None else -1, x.stride(0), dy.stride(0),
                             R, Cin, Cout, co_begin, co_count, slab_rows))
                tiles[taps].append((R * taps, jid, -(-co_count // 128), -(-Cin // CI_TILE[taps]), S))
            v = conv.weight_v if conv.weight_norm else conv.weight
            wnbs.append((part_off, pb_off, v.data_ptr(), conv.weight_g.data_ptr() if dg is not None else 0,
                         pc.inv_norm.data_ptr() if dg is not None else 0, dv.data_ptr(), dg.data_ptr() if dg is not None else 0,
                         db.data_ptr() if db is not None else 0, S, Cout, Cin, taps, row, int(self.accumulate), 0, 0))
            row += Cout
            max_n = max(max_n, Cin * taps)
        return jobs, tiles, wnbs, row, max_n, off

    def flush(self):
        self._flush_ln()
        if not self.items:
            return
        self._flush_convs()
        cb = getattr(self.site, "_gt_after_flush", None) if self.site is not None else None
        if cb is not None and not (ASYNC or self.force_side):
            cb(self.params_done)
        self.params_done = []

    def _flush_convs(self):
        dev = self.dev
        if (ASYNC or self.force_side) and dev.type == "cuda":
            cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
            side.wait_stream(cur)
            for conv, R, parts, dv, dg, db in self.items:          # these outlive the backward node on the side stream
                for x, dy, _, _ in parts:
                    x.record_stream(side); dy.record_stream(side)
                for t in (dv, dg, db):
                    if t is not None:
                        t.record_stream(side)
            with torch.cuda.stream(side):
                self._flush()
            _PENDING.add(str(dev))
        else:
            self._flush()

    def _keep(self, obj):
        """Whatever a captured graph reads (pinned staging, device tables from the graph's private pool) must outlive
        the graph — and go with it: under a trainer's capture it joins that capture's CaptureKeep; a capture nobody owns hangs it
        on the module that owns the queue, so that it dies with the model instead of at interpreter shutdown (device memory of a
        graph pool freed after the allocator is gone aborts)."""
        if capture_keep(obj):
            return
        if self.site is not None:
            lst = getattr(self.site, "_wgrad_keep", None)
            if lst is None:
                lst = []
                object.__setattr__(self.site, "_wgrad_keep", lst)
            lst.append(obj)
        else:
            _KEEP.append(obj)

    def _flush(self):
        L = _lib.lib()
        dev = self.dev
        jobs, tiles, wnbs, rows, max_n, nbytes = self._plan()
        ws = _scratch(dev, nbytes)
        base = ws.data_ptr()
        capturing = torch.cuda.is_current_stream_capturing()

        def upload(arr):
            raw = arr.view(np.uint8).reshape(-1)
            if capturing:
                # pinned allocation is illegal while a stream is capturing: take a buffer set aside earlier; the
                # graph's copy node reads it at every replay, so it is never reused
                if not _POOL or raw.size > _POOL_BYTES:
                    raise RuntimeError("wgrad tables: no pinned staging buffer available inside graph capture "
                                       "(run one eager step first)")
                full = _POOL.pop()
                host = full[: raw.size]
                host.numpy()[:] = raw
                self._keep(full)                         # the whole buffer: it returns to the pool when the capture is evicted
                # The tables of a captured step never change (the graph's buffers are static).  With a table arena open (the
                # trainer allocates it BEFORE the capture, outside the graph's memory pool — a pool block would be reused by other
                # tensors of the step and clobbered at every replay) they are copied ONCE, right after the capture
                # (sync_uploads), instead of by a memcpy node that every replay runs (12 of them per step at cfg 2).
                ar = _TABLE_ARENA.get(str(dev))
                nb = (raw.size + 255) & ~255
                if ar is not None and ar["off"] + nb <= ar["buf"].numel():
                    dst = ar["buf"][ar["off"]:ar["off"] + raw.size]
                    ar["off"] += nb
                    _PENDING_UPLOADS.append((dst, host))
                    return dst
                return host.to(dev, non_blocking=True)
            host = torch.from_numpy(raw).pin_memory()
            return host.to(dev, non_blocking=True)

        if capturing:
            cache = {}                                   # tables of a captured step belong to the graph alone
        else:
            _fill_pool()
            cache = getattr(self.site, "_wgrad_tables", None) if self.site is not None else None
            if cache is None:
                cache = {}
                if self.site is not None:
                    object.__setattr__(self.site, "_wgrad_tables", cache)
        # tiles depend on shapes only
        skey = tuple((t, tuple(x[2:] for x in tiles[t]), tuple(j[6:] for j in jobs)) for t in (5, 3, 1))
        if cache.get("skey") != skey:
            tl, counts = [], []
            for taps in (5, 3, 1):
                n = 0
                order = sorted(range(len(tiles[taps])), key=lambda i: -tiles[taps][i][0])       # heavy tiles first
                # XCD-aware order: workgroup i of a launch lands on XCD i % 8 (each XCD has its own L2).  The nco x nci tiles of one
                # (job, slab) read the same dY / X rows — nci tiles share every dY block, nco tiles every X block — so they are dealt
                # to ONE XCD (consecutive slots of its queue); dealt round-robin, each of them pulled its own copy over the fabric
                # (3x the bytes for the decoder's 384 x 192 convs).
                queues = [[] for _ in range(XCDS)]
                for i in order:
                    _, jid, nco, nci, S = tiles[taps][i]
                    for sl in range(S):
                        q = min(queues, key=len)
                        q += [(jid, co * 128, ci * CI_TILE[taps], sl) for co in range(nco) for ci in range(nci)]
                flat = [q[pos] for pos in range(max(map(len, queues), default=0)) for q in queues if pos < len(q)]
                if flat:
                    t = np.zeros(len(flat), dtype=TILE)
                    a = np.array(flat, dtype=np.int64)
                    t["job"], t["co0"], t["ci0"], t["slab"] = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
                    tl.append(t)
                    n = len(flat)
                counts.append(n)
            ta = np.concatenate(tl) if tl else np.zeros(1, dtype=TILE)
            cache["skey"], cache["tiles"], cache["counts"] = skey, upload(ta), counts
            cache["pkey"] = None
        pkey = (base, tuple(j[:4] for j in jobs), tuple(w[2:8] for w in wnbs), self.accumulate)
        if cache.get("pkey") != pkey:
            ja = np.array([(j[0], j[1], base + j[2], (base + j[3]) if j[3] >= 0 else 0) + j[4:] for j in jobs], dtype=JOB)
            wa = np.array([(base + w[0], base + w[1]) + w[2:] for w in wnbs], dtype=WNB)
            cache["pkey"], cache["jobs"], cache["wnb"] = pkey, upload(ja), upload(wa)
        counts = cache["counts"]
        st = _lib.current_stream(dev)
        assert all(L.gt_conv_wgrad_ci_tile(t) == w for t, w in CI_TILE.items()), "wgrad.CI_TILE does not match the library's tile widths"
        _lib.check(L.gt_conv_wgrad_batched(_lib.ptr(cache["jobs"]), _lib.ptr(cache["tiles"]), counts[0], counts[1], counts[2], st),
                   "gt_conv_wgrad_batched")
        _lib.check(L.gt_weightnorm_bwd_batched(_lib.ptr(cache["wnb"]), len(wnbs), rows, max_n, st), "gt_weightnorm_bwd_batched")
        if capturing:
            self._keep(cache)
        self.items = []
