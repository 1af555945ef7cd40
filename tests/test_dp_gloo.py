"""CPU test of the data-parallel gradient exchange (world_size 2, gloo): bucketed flat all-reduce of
glow_tts_amd.train.GradBuckets == mean of the per-rank gradients == gradient of the concatenated
batch's mean loss (what DDP gives the reference, train_ms_emo_lang_pitch.py:166)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glow_tts_amd.train import GradBuckets
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s)) for s in [(7, 5), (3,), (11, 2, 3), (1,), (64, 33)]]
    x = torch.randn(4, 5, generator=torch.Generator().manual_seed(100 + rank))
    loss = sum((p * (rank + 1)).sum() * 0.1 for p in params[1:]) + (params[0] @ x.t()).pow(2).mean()
    loss.backward()
    local = [p.grad.clone() for p in params]
    gb = GradBuckets(params, world, bucket_mb=1e-4)            # tiny buckets -> several collectives
    assert len(gb.buckets) > 1
    # the data-parallel trainer reduces the flat buffer in two ranges (decoder tail first, then the head): same result
    split = gb.offsets[2]
    gb.gather()
    gb.allreduce(split, None, wait=False)
    gb.allreduce(0, split, wait=True)
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, lst in zip(local, gathered):
        dist.all_gather(lst, g)
    ok = all(torch.allclose(p.grad, torch.stack(lst).mean(0), atol=1e-6) for p, lst in zip(params, gathered))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert sorted(res) == [(0, True), (1, True)]


def _worker_bf16(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from glow_tts_amd.train import GradBuckets
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(s)) for s in [(7, 5), (3,), (11, 2, 3), (1,), (64, 33)]]
    gen = torch.Generator().manual_seed(100 + rank)
    for p in params:
        p.grad = torch.randn(p.shape, generator=gen) * (10.0 ** float(torch.randint(-3, 3, (1,), generator=gen)))
    local = [p.grad.clone() for p in params]
    gb = GradBuckets(params, world, bucket_mb=1e-3, wire="bf16")
    assert len(gb.buckets) > 1
    split = gb.offsets[2]
    gb.gather()
    gb.allreduce(split, None, wait=False)
    gb.allreduce(0, split, wait=True)
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, lst in zip(local, gathered):
        dist.all_gather(lst, g)
    # bf16 on the wire twice (8 mantissa bits), fp32 accumulation: relative error of the mean <= 2 * 2^-8 of the largest term
    ok = all(((p.grad - torch.stack(lst).mean(0)).abs() <= 2.0 ** -7 * torch.stack(lst).abs().max(0).values + 1e-30).all()
             for p, lst in zip(params, gathered))
    # ... and the ranks agree bit for bit (every rank gathers the same reduced shards)
    flat = gb.flat.clone()
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    same = all(torch.equal(both[0], b) for b in both[1:])
    q.put((rank, bool(ok), bool(same)))
    dist.destroy_process_group()


def test_bf16_wire_allreduce_world2():
    """GradBuckets(wire="bf16"): all-to-all of bf16 shard copies, fp32 sum, all-gather of the reduced bf16 shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(timeout=60) for p in procs]
    assert sorted(res) == [(0, True, True), (1, True, True)]
