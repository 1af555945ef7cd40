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

import torch
import torch.distributed as dist

from . import models

BASE_MODEL = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1,
                  n_blocks_dec=12, n_layers_enc=6, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5,
                  n_block_layers=4, n_sqz=2, prenet=True, mean_only=True, hidden_channels_enc=192,
                  hidden_channels_dec=192, window_size=4)      # == reference configs/base.json "model"


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


class GradBuckets:
    """Flat gradient buckets + asynchronous all-reduce (mean) on a communication stream."""

    def __init__(self, params, world, bucket_mb=64):
        self.world = world
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, cur_n = [], [], 0
        cap = bucket_mb * (1 << 20) // 4
        for p in reversed(self.params):                      # roughly the order gradients become ready
            cur.append(p); cur_n += p.numel()
            if cur_n >= cap:
                self.buckets.append(cur); cur, cur_n = [], 0
        if cur:
            self.buckets.append(cur)
        dev = self.params[0].device
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=dev) for b in self.buckets]
        self.on_gpu = dev.type == "cuda"
        self.comm = torch.cuda.Stream(device=dev) if (world > 1 and self.on_gpu) else None

    def reduce_all(self):
        """Pack every bucket and all-reduce it; bucket i+1 is packed while bucket i is on the wire."""
        if self.world == 1:
            return
        cur = torch.cuda.current_stream() if self.on_gpu else None
        for b, flat in zip(self.buckets, self.flat):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b]
            views = list(torch.split(flat, [p.numel() for p in b]))
            torch._foreach_copy_([v.view_as(g) for v, g in zip(views, grads)], grads)
            if self.on_gpu:
                self.comm.wait_stream(cur)
                with torch.cuda.stream(self.comm):
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG)       # RCCL over xGMI
            else:                                                       # gloo (CPU tests): SUM then scale
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.mul_(1.0 / self.world)
        if self.on_gpu:
            cur.wait_stream(self.comm)
        for b, flat in zip(self.buckets, self.flat):
            views = torch.split(flat, [p.numel() for p in b])
            for p, v in zip(b, views):
                p.grad = v.view_as(p)


class Trainer:
    """zero_grad -> forward -> loss -> backward -> all-reduce -> grad-norm -> AdamW step.

    graph=True captures the whole step (forward, MAS, backward, optimizer: ~1.5 k kernel launches) into
    ONE HIP graph after three eager warm-up steps and replays it; the batch then lives in static
    buffers (`step` copies into them) and dropout masks still change every replay because every
    dropout kernel mixes the device-resident seed word (ops.seed_word) that the graph itself bumps."""

    def __init__(self, model, lr=2e-4, betas=(0.9, 0.98), eps=1e-9, world=1, graph=False):
        self.model = model
        self.world = world
        self.graph_mode = bool(graph) and world == 1
        self.opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=betas, eps=eps, fused=True, capturable=self.graph_mode)
        self.buckets = GradBuckets(list(model.parameters()), world)
        self.grad_norm = None
        self._graph = None
        self._static = None
        self._out = None

    def _step_impl(self, ids, t_x, y, t_y):
        from . import ops
        m = self.model
        ops.bump_seed(ids.device)
        self.opt.zero_grad(set_to_none=True)
        (z, z_m, z_logs, logdet, z_mask), _, (attn, l_length, _, _), _, _ = m(ids, t_x, y, t_y)
        l_mle = models.mle_loss(z, z_m, None if m.mean_only else z_logs, logdet, z_mask)
        loss = l_mle + torch.sum(l_length)
        loss.backward()
        self.buckets.reduce_all()
        grads = [p.grad for p in m.parameters() if p.grad is not None]
        # reference commons.clip_grad_value_(params, None): total grad norm, no clipping (one device
        # reduction instead of ~1.8k .item() syncs)
        self.grad_norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
        self.opt.step()
        return loss.detach(), l_mle.detach()

    def _capture(self, ids, t_x, y, t_y):
        self._static = [t.clone() for t in (ids, t_x, y, t_y)]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):                       # warm-up: one-time attribute calls, scratch growth, optimizer state
                self._step_impl(*self._static)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self._step_impl(*self._static)

    def step(self, ids, t_x, y, t_y):
        if not self.graph_mode:
            return self._step_impl(ids, t_x, y, t_y)
        if self._graph is None:
            self._capture(ids, t_x, y, t_y)
        else:
            for dst, src in zip(self._static, (ids, t_x, y, t_y)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self._graph.replay()
        return self._out
