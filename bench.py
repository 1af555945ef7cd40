#!/usr/bin/env python3
"""bench.py — Glow-TTS training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5dec|cfg5]

One "step" = one full training step of the hot path (zero_grad, TextEncoder + FlowSpecDecoder forward, logp + MAS,
mle + duration loss [+ pitch / energy predictor losses for cfg 5], backward, gradient all-reduce, grad-norm, AdamW) on one
synthetic LJSpeech-shaped batch resident in HBM (seed 1234 + rank, random-init weights).  The timed region
ROTATES over 8 distinct seeded batches (different lengths, several ragged-row buckets), so graph-bucket switches and the
per-batch refresh of the row contexts are inside it; their graphs are captured once, up front and untimed (a capture is
set-up cost, like compilation: one per bucket per run).  With N > 1 the driver launches this file under
torch.distributed.run, one rank per GPU: the batch is sharded by utterance (weak scaling, B per GPU fixed) and the only
collective is the RCCL gradient all-reduce.

Prints ONE JSON line on rank 0:
  value            = valid mel-frames / s over all ranks (BASELINE.json metric, first half); wall clock of exactly K steps
  step_ms_median   = median of the K per-step times (HIP events recorded on the step's stream, read after the region)
  mas              = MAS alignments / s of gt_mas_f32 alone on the step's own lattice shape (second half) + its CPU baselines
                     (reference Cython core on 1 core — what the reference uses — and the C restatement on all cores)
  roofline         = the dominant kernel family by time, the whole-WaveNet forward kernel (gt_wn_stack_fwd: 4 x (k=5 conv + gate) +
                     3 x residual 1x1 in one launch; the per-layer gt_wn_layer_fwd where a batch's rows take that path):
                     algorithmic FLOPs / its duration INSIDE the replayed graph (device-side begin / end stamps of every one of
                     its launches in the timed steps; `profiles/` holds the rocprofv3 summary of the same command), plus
                     `step` = whole-step FLOP/s over the bf16 MFMA peak
  cpu_baseline     = the oracle's training step (PyTorch-CPU fp32 restatement of the reference, oracle/glowtts_ref.py +
                     reference Cython MAS from oracle/_ref) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0
N_BATCHES = 8                     # distinct batches the timed region rotates over

CFG5_MODEL = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1, n_blocks_dec=12,
                  n_layers_enc=10, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5, n_block_layers=4, n_sqz=2,
                  prenet=True, mean_only=True, hidden_channels_enc=192, hidden_channels_dec=192, window_size=4, gin_channels=512,
                  use_sdp=True, use_spk_embeds=True, use_lang_embeds=True, use_emo_embeds=True, lin_channels=4, emoin_channels=1024,
                  use_spp=True, use_sep=True)     # == reference configs/base_blank_emo_lang_pitch.json "model"

WORKLOADS = {
    "cfg2": dict(B=32, T_x=150, T_y=800, desc="configs/base.json, LJSpeech-shaped synthetic batches, B=32/GPU, T_x<=150, T_y<=800, bf16 GEMMs"),
    "cfg3": dict(B=32, T_x=375, T_y=872, desc="configs/base_blank.json-shaped synthetic batches, B=32/GPU, T_x<=375, T_y<=872, bf16 GEMMs"),
    "cfg4": dict(B=20, T_x=235, T_y=500, gin=256,
                 desc="configs/base_blank_ms.json-shaped synthetic batches (multi-speaker, gin_channels=256, g ~ N(0,1) [B,256,1]), "
                      "B=20/GPU, T_x<=235, T_y<=500, bf16 GEMMs"),
    "cfg5dec": dict(B=32, T_x=127, T_y=400, gin=512, n_layers_enc=10, prosody=True, n_lang=10, lin=4,
                    desc="cfg 5's decoder side only (round 1's workload, kept for comparison): gin 512, 10 encoder layers, language "
                         "embedding, 3 WaveNets per coupling block; no emotion front end, deterministic duration predictor"),
    "cfg5": dict(B=32, T_x=127, T_y=400, full=True,
                 desc="configs/base_blank_emo_lang_pitch.json as the reference runs it: FlowGenerator(**hps.model) with the speaker / "
                      "emotion front end, StochasticDurationPredictor, stochastic pitch / energy predictors, 3 WaveNets per coupling "
                      "block, 10 encoder layers; B=32/GPU, T_x<=127, T_y<=400, bf16 GEMMs (bf16x3 in the predictors)"),
}


def make_batch(wl, rank, dev, i):
    """Batch i of the rotation (SURVEY §8d shapes; batch 0 is the seed-1234 batch of round 1's bench) + its conditioning."""
    from glow_tts_amd import train
    B, gin = wl["B"], wl.get("gin", 0)
    ids, t_x, y, t_y = train.synth_batch(B, wl["T_x"], wl["T_y"], rank + 1000 * i, dev, n_vocab=187 if wl.get("full") else 148)
    g = torch.Generator().manual_seed(4321 + rank + 1000 * i)
    cond = {}
    if wl.get("full"):                               # cfg 5: g ~ N(0,1)[B,512], emo ~ U{0..4}, emo_cartesian ~ U[0,1.5)^3, pitch, energy, l
        cond["g"] = torch.randn(B, 512, generator=g).to(dev)
        cond["emo"] = torch.randint(0, 5, (B,), generator=g).to(dev)
        cond["emo_cartesian"] = (torch.rand(B, 3, generator=g) * 1.5).to(dev)
    elif gin:
        cond["g"] = torch.randn(B, gin, 1, generator=g).to(dev)
    if wl.get("prosody") or wl.get("full"):          # pitch ~ U[80,280) Hz with 30 % unvoiced zeros, energy ~ U[1,11)
        cond["pitch"] = ((80 + 200 * torch.rand(B, 1, wl["T_y"], generator=g)) * (torch.rand(B, 1, wl["T_y"], generator=g) > 0.3)).to(dev)
        cond["energy"] = (1 + 10 * torch.rand(B, 1, wl["T_y"], generator=g)).to(dev)
        cond["l"] = torch.randint(0, 3, (B,), generator=g).to(dev)
    return dict(ids=ids, t_x=t_x, y=y, t_y=t_y, lh=(t_x.tolist(), t_y.tolist()), cond=cond, valid=int(t_y.sum().item()))


def mas_leg(dev, wl, rank, iters=100):
    """MAS alone on a lattice of the workload's shape: alignments/s and achieved HBM GB/s on the device, and the same lattice
    through the reference's Cython core (1 core) / the C restatement on all host cores."""
    from glow_tts_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(1234 + rank)
    B, T_x, T_y = wl["B"], wl["T_x"], wl["T_y"]
    t_x = torch.randint(max(1, int(T_x * 0.4)), T_x + 1, (B,), generator=g, dtype=torch.int32)
    t_y = torch.randint(max(1, T_y * 3 // 16), T_y // 2 + 1, (B,), generator=g, dtype=torch.int32) * 2
    t_y = torch.maximum(t_y, t_x + (t_x % 2))
    t_x[0], t_y[0] = T_x, T_y
    logp_cpu = torch.randn(B, T_x, T_y, generator=g) * 5.0 - 100.0
    logp = logp_cpu.to(dev)
    t_xd, t_yd = t_x.to(dev), t_y.to(dev)
    path = torch.empty_like(logp)
    ws_bytes = L.gt_mas_workspace_bytes(B, T_x, T_y)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def run():
        rc = L.gt_mas_f32(_lib.ptr(logp), None, _lib.ptr(t_xd), _lib.ptr(t_yd), _lib.ptr(path), _lib.GT_DT_F32, None, None,
                          B, T_x, T_y, logp.stride(0), logp.stride(1), _lib.ptr(ws), ws_bytes, None, _lib.current_stream(dev))
        assert rc == 0, rc
    for _ in range(10):
        run()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    assert int(path.sum().item()) == int(t_y.sum().item())
    algo_bytes = 8.0 * B * T_x * T_y + 8.0 * B
    out = {"alignments_per_sec": B / (ms * 1e-3), "ms_per_batch": ms, "achieved_GBps": algo_bytes / (ms * 1e-3) / 1e9,
           "hbm_frac": algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo_bytes}
    if rank == 0:
        out["cpu"] = mas_cpu(logp_cpu.numpy(), t_x.numpy(), t_y.numpy(), path.cpu().numpy())
    return out


def mas_cpu(value, t_x, t_y, want_path):
    """The same lattice on the host: reference core.pyx compiled by oracle/Makefile (oracle/_ref) on ONE core — the reference
    builds it without OpenMP, so its prange is serial — and the C restatement with one utterance per thread."""
    from oracle import mas as omas
    B = value.shape[0]
    use_ref = omas.ref_module() is not None
    core = omas.ref_maximum_path_c if use_ref else omas.oracle_maximum_path_c
    times = []
    for _ in range(5):
        v = np.ascontiguousarray(value.copy()); p = np.zeros(v.shape, dtype=np.int32)
        t0 = time.perf_counter()
        core(p, v, t_x, t_y)
        times.append(time.perf_counter() - t0)
    assert np.array_equal(p, want_path.astype(np.int32)), "device MAS path != CPU path on the bench lattice"
    one = min(times)
    # all cores: the C restatement's batch loop under OpenMP, one utterance per thread (a Python thread pool over 50-us ctypes
    # calls measured the dispatch overhead, not the cores: round 2's "all cores" figure was below the one-core one)
    best, nthr = None, 1
    p = np.empty(value.shape, dtype=np.int32)
    for want in sorted({min(os.cpu_count() or 1, B), min(16, B), min(8, B)}, reverse=True):   # the box's share of its host cores varies
        for _ in range(3):
            v = np.ascontiguousarray(value.copy()); p.fill(0)
            t0 = time.perf_counter()
            used = omas.oracle_maximum_path_omp(p, v, t_x, t_y, want)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, nthr = dt, used
    assert np.array_equal(p, want_path.astype(np.int32)), "OpenMP batch loop != device path"
    return {"one_core": {"alignments_per_sec": B / one, "kind": "reference" if use_ref else "port",
                         "what": ("reference monotonic_align/core.pyx (oracle/_ref), serial as the reference builds it" if use_ref
                                  else "C restatement oracle/mas_oracle.c"), "cores": 1},
            "all_cores": {"alignments_per_sec": B / best, "kind": "port", "what": "oracle/mas_oracle.c batch loop under OpenMP, one utterance per thread",
                          "cores": nthr}}


def cpu_baseline(batch, model, wl, budget_utts=8):
    """The oracle's training step on the host cores: same weights, a bounded sample of batch 0."""
    from oracle import glowtts_ref as R
    from oracle import mas as omas
    mas_core = omas.ref_maximum_path_c if omas.ref_module() is not None else omas.oracle_maximum_path_c

    def mp(logp, mask):
        p = omas.oracle_maximum_path(logp.numpy(), mask.numpy(), core=mas_core)
        return torch.from_numpy(p).float()
    ids, t_x, y, t_y, cond = batch["ids"], batch["t_x"], batch["y"], batch["t_y"], batch["cond"]
    n = min(budget_utts, ids.shape[0])
    P = {k: v.detach().cpu().float().clone().requires_grad_(v.dtype.is_floating_point and "bins" not in k) for k, v in model.state_dict().items()}
    ids_c, tx_c, y_c, ty_c = ids[:n].cpu(), t_x[:n].cpu().long(), y[:n].cpu(), t_y[:n].cpu().long()
    Tx, Ty = int(tx_c.max()), int(ty_c.max())
    ids_c, y_c = ids_c[:, :Tx], y_c[:, :, :Ty]
    c = {k: v[:n].cpu() for k, v in cond.items()}
    if "pitch" in c:
        c["pitch"], c["energy"] = c["pitch"][:, :, :Ty], c["energy"][:, :, :Ty]

    def one_step(it, n):
        if wl.get("full"):
            g = torch.Generator().manual_seed(it)
            noises = (torch.randn(n, 2, Tx, generator=g), torch.randn(n, 1, Ty, generator=g), torch.randn(n, 1, Ty, generator=g))
            out = R.train_forward_full(P, ids_c[:n], tx_c[:n], y_c[:n], ty_c[:n], mp, CFG5_MODEL, c["g"][:n], c["emo"][:n], c["emo_cartesian"][:n],
                                       c["pitch"][:n], c["energy"][:n], c["l"][:n], noises)
        else:
            hp = dict(hidden_channels=192, n_layers_enc=wl.get("n_layers_enc", 6), n_heads=2, window_size=4, kernel_size=3, prenet=True,
                      mean_only=True, n_blocks_dec=12, n_block_layers=4, kernel_size_dec=5, n_sqz=2)
            sl = lambda k: None if k not in c else c[k][:n]                           # noqa: E731
            out = R.train_forward(P, ids_c[:n], tx_c[:n], y_c[:n], ty_c[:n], mp, hp, g=sl("g"), pitch=sl("pitch"), energy=sl("energy"),
                                  l=None if "l" not in c else torch.nn.functional.embedding(c["l"][:n], P["emb_l.weight"]).unsqueeze(-1))
        out["loss"].backward()
        for v in P.values():
            v.grad = None

    def timed(it, n):
        t0 = time.perf_counter()
        one_step(it, n)
        return time.perf_counter() - t0
    # the thread count PyTorch-CPU does best with on this box (its default, every hardware thread, is far from it: 128 threads ran the
    # step 30 x slower than 8 on the round-3 box): probed on 2 utterances, then the full sample is timed three times with the winner
    all_threads = torch.get_num_threads()
    probe = {}
    for k in sorted({all_threads, min(all_threads, 32), min(all_threads, 16), min(all_threads, 8)}):
        torch.set_num_threads(k)
        timed(0, min(2, n))                                                            # warm
        probe[k] = timed(1, min(2, n))
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    times = [timed(it, n) for it in range(3)]
    torch.set_num_threads(all_threads)
    dt = min(times)
    return {"value": float(ty_c.sum()) / dt, "unit": "mel-frames/s", "cores": cores, "kind": "port",
            "sample": f"fwd+loss+bwd (no optimizer) of the first {n} utterances of batch 0 ({int(ty_c.sum())} valid frames), "
                      f"PyTorch-CPU fp32 oracle + {'reference Cython' if mas_core is omas.ref_maximum_path_c else 'C port'} MAS, best of 3 at the best of {sorted(probe)} threads",
            "s_per_step": dt, "threads_probe_s": {str(k): v for k, v in probe.items()}}


def step_flops(wl, batch):
    """Algorithmic FLOPs of one training step on this batch (3 x forward): decoder 21.35 (62.9 with the three WaveNets
    of cfg 5) MFLOP per valid mel frame + encoder ~15.0 (23.8 with 10 layers) MFLOP per valid token + logp 2*80 MAC per lattice cell."""
    dec = 62.9e6 if (wl.get("prosody") or wl.get("full")) else 21.35e6
    enc = 23.8e6 if (wl.get("n_layers_enc", 6) == 10 or wl.get("full")) else 15.0e6
    fr, tok = float(batch["valid"]), float(sum(batch["lh"][0]))
    cells = float(sum(a * b for a, b in zip(*batch["lh"])))
    return 3.0 * (dec * fr + enc * tok) + 2.0 * 2.0 * 80.0 * cells


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--late-adam", action="store_true", help="dev: the optimizer's pass over the decoder's parameters after the whole backward "
                    "(Trainer(early_decoder_adam=False)) instead of right behind the decoder's weight gradients")
    ap.add_argument("--head-pack", action="store_true", help="dev: pack all weights at the head of the step (Trainer(pack_in_tail=False)) "
                    "instead of the decoder's at the end of the previous one")
    ap.add_argument("--join-roots", action="store_true", help="dev (cfg 5): join the predictors' branch at the end of the forward and run "
                    "ONE backward root (Trainer(split_roots=False)) instead of one root per stream")
    ap.add_argument("--split-graph", type=int, default=None, choices=[0, 1],
                    help="dev: force (1) / forbid (0) the phased, three-graph step (Trainer(split_graph=)); at N = 1 it shows what "
                         "that schedule costs without any collective (6.0 vs 4.84 ms: why one backward is the default at every N)")
    ap.add_argument("--marks", action="store_true", help="dev: print the device-clock timeline of the last replayed step (ops.Marks) to stderr")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying HIP graphs")
    ap.add_argument("--one-batch", action="store_true", help="round 1's protocol: the same batch every step")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"],
                    help="N > 1: gradient exchange in fp32 (the reference's DDP all-reduce; default) or bf16 on the wire with fp32 accumulation")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from glow_tts_amd import _lib, train
    _lib.lib()                                       # fails loudly if the HIP library is missing
    wl = WORKLOADS[args.workload]
    torch.manual_seed(1234)                          # identical initial weights on every rank
    if wl.get("full"):
        model = train.build_model(dict(CFG5_MODEL, n_lang=10), n_vocab=187, device=dev).train()
    else:
        gin = wl.get("gin", 0)
        cfg = dict(train.BASE_MODEL, gin_channels=gin, n_layers_enc=wl.get("n_layers_enc", 6), with_prosody_wn=bool(wl.get("prosody")),
                   n_lang=wl.get("n_lang", 0), lin_channels=wl.get("lin", 0))
        model = train.build_model(cfg if (gin or wl.get("prosody")) else None, device=dev).train()
    if world > 1:
        for p in model.parameters():
            torch.distributed.broadcast(p.data, 0)
    use_graph = not args.no_graph
    tr = train.Trainer(model, world=world, graph=use_graph, kernel_stamps=True, grad_wire=args.grad_wire,
                       split_graph=None if args.split_graph is None else bool(args.split_graph),
                       early_decoder_adam=not args.late_adam, pack_in_tail=not args.head_pack, split_roots=not args.join_roots)
    nb = 1 if args.one_batch else N_BATCHES
    batches = [make_batch(wl, rank, dev, i) for i in range(nb)]
    call = lambda b: tr.step(b["ids"], b["t_x"], b["y"], b["t_y"], lengths_host=b["lh"], **b["cond"])      # noqa: E731

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    # graphs of every row bucket the rotation meets: captured up front, untimed, without taking a training step
    if args.marks:
        from importlib import import_module
        _ops = import_module(train.__name__.rsplit(".", 1)[0] + ".ops")
        _ops.MARKS = _ops.Marks(dev)
    if tr.stamps is not None:                       # every step (eager warm-ups of the captures included) takes a row of stamp slots
        assert args.steps + args.warmup + 3 * len(batches) + 8 <= tr.stamps.max_steps, "raise KernelStamps(max_steps=) for this many steps"
    t_cap = time.perf_counter()
    tr.precapture([(b["ids"], b["t_x"], b["y"], b["t_y"], b["lh"], b["cond"]) for b in batches])
    torch.cuda.synchronize(dev)
    t_cap = time.perf_counter() - t_cap
    for i in range(args.warmup):
        call(batches[i % nb])
    barrier()
    steps0 = None if tr.stamps is None else int(tr.stamps.base.item()) // tr.stamps.per_step
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    stream = torch.cuda.current_stream(dev)
    valid_total = 0
    t0 = time.perf_counter()
    evs[0].record(stream)
    for i in range(args.steps):
        b = batches[i % nb]
        loss, mle = call(b)
        evs[i + 1].record(stream)
        valid_total += b["valid"]
    barrier()
    wall = time.perf_counter() - t0
    step_ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))
    if world > 1:
        t = torch.tensor([wall, float(valid_total)], device=dev, dtype=torch.float64)
        tmax = t.clone(); torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tsum = t.clone(); torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        wall, total_valid = float(tmax[0]), float(tsum[1])
    else:
        total_valid = float(valid_total)
    assert torch.isfinite(loss).item(), "training step diverged"
    if args.marks and rank == 0:
        for us, name in _ops.MARKS.report():
            print(f"  {us:9.1f} us  {name}", file=sys.stderr)

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        flops_step = sum(step_flops(wl, batches[i % nb]) for i in range(args.steps)) / args.steps
        rows_keys = sorted({tr._rows_key(-(-b["ids"].shape[1] // tr.pad_tx) * tr.pad_tx, -(-b["y"].shape[2] // tr.pad_ty) * tr.pad_ty, b["lh"])
                            for b in batches})
        line = {
            "metric": "mel_frames_per_sec_train_step",
            "value": total_valid / wall,
            "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["desc"], "batch_per_gpu": wl["B"], "T_x": wl["T_x"], "T_y": wl["T_y"],
                       "batches_rotated": nb, "valid_frames_per_gpu_step_mean": valid_total / args.steps,
                       "padded_frames_per_gpu_step": wl["B"] * wl["T_y"],
                       "rows_layout": "ragged" if tr.cfg.ragged else "uniform",
                       "row_buckets_(text, mel, frames)": [list(k) for k in rows_keys], "graphs_captured": tr.n_captures,
                       "capture_s_untimed": round(t_cap, 2),
                       "parallelism": f"dp{world} (utterance-sharded, RCCL gradient all-reduce, {args.grad_wire} on the wire)",
                       "launch": (("one HIP graph per step (one per ragged-row bucket, captured up front)" if world == 1 else
                                   ("three HIP graphs per step (forward + decoder-side backward | encoder backward | optimizer), the RCCL "
                                    "all-reduces of the flat gradient buffer launched between them" if tr.split else
                                    "two HIP graphs per step (forward + backward | optimizer) with the RCCL all-reduce of the flat gradient "
                                    "buffer between them")) if tr.graph_mode else "eager launches"),
                       "sub_graph": ("the reference's own FlowGenerator.forward for this config (models.py:1007-1133)" if wl.get("full") else
                                     "upstream-equivalent live sub-graph of the base configs (one WN per coupling block, deterministic "
                                     "DurationPredictor) — SURVEY F1/F2/F4: the fork's class does not construct for them"),
                       "final_loss": float(loss)},
            "step_ms_median": step_ms[len(step_ms) // 2], "step_ms_min": step_ms[0], "step_ms_max": step_ms[-1],
            "padded_frames_per_sec": world * wl["B"] * wl["T_y"] / (wall / args.steps),
        }
        # ---- roofline: the fused WaveNet-layer forward kernel, timed INSIDE the replayed graphs by its own device stamps
        roof = {"bound": "mfma", "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None}
        if tr.stamps is not None:
            d = tr.stamps.durations_ticks()[steps0:steps0 + args.steps].double()          # [K, slots per step] in 10-ns ticks
            nl = model.decoder.n_layers
            rows_valid = (valid_total / args.steps) / 2.0                                 # squeezed frames
            # which kernel stamped which slot of a step (ops.KernelStamps.kinds): the whole-WaveNet launches (csrc/wn_stack.hip), or —
            # for row counts whose extra workgroups would not fit one round of CUs — the per-layer launches (csrc/wn_layer.hip).
            # The family with the larger share of the stamped time is the roofline kernel; slots nobody wrote in a step stay 0.
            kinds = tr.stamps.kinds
            fam = {"stack": [k for k, v in kinds.items() if v == {f"stack{nl}"}], "layer": [k for k, v in kinds.items() if v == {"layer"}],
                   "layer_last": [k for k, v in kinds.items() if v == {"layer_last"}]}
            mixed = sorted(k for k, v in kinds.items() if len(v) > 1)
            tot = {f: float(d[:, sl].sum()) if sl else 0.0 for f, sl in fam.items()}
            if tot["stack"] >= tot["layer"] + tot["layer_last"]:
                used = d[:, fam["stack"]]
                used = used[used > 0]
                us = used.mean().item() / 100.0                                           # wall_clock64: 100 MHz
                flops_launch = 2.0 * rows_valid * (nl * 384 * 192 * 5 + (nl - 1) * 192 * 192)
                name = (f"gt_wn_stack_fwd_kernel (a whole WaveNet forward in one launch: {nl} x (k=5 conv 192->384 + gate) + {nl - 1} x "
                        f"residual 1x1, halo recomputed per 52-row tile; {used.numel() / args.steps:.1f} launches per step)")
                extra, n_l = {}, used.numel()
                pmc_name = "r03_decoder_pmc.json"
            else:
                used = d[:, fam["layer"]]
                used = used[used > 0]
                us = used.mean().item() / 100.0
                flops_launch = 2.0 * rows_valid * (384 * 192 * 5 + 192 * 192)
                name = ("gt_wn_layer_fwd_kernel<true> (WaveNet layer: k=5 conv 192->384 + gate + residual 1x1, one launch; "
                        f"{used.numel() / args.steps:.1f} launches per step)")
                last = d[:, fam["layer_last"]]
                extra, n_l = {"launch_us_last_layer_variant": last[last > 0].mean().item() / 100.0 if fam["layer_last"] else None}, used.numel()
                pmc_name = "r02_wn_layer_pmc.json"
            extra["stamped_us_per_step"] = {f: v / 100.0 / args.steps for f, v in tot.items()}
            if mixed:
                extra["slots_with_both_kernel_families"] = mixed                          # (different row buckets routed differently)
            tf = flops_launch / (us * 1e-6) / 1e12
            roof.update({"achieved": tf, "frac": tf / MFMA_BF16_PEAK_TFLOPS, "kernel": name,
                         "algorithmic_flops_per_launch": flops_launch, "launch_us": us, **extra,
                         "how": "device-side begin / end stamps (wall_clock64, 100 MHz) written by every launch of the kernel inside the "
                                "replayed HIP graphs of the timed steps: max(end) - min(start) per launch, mean over "
                                f"{n_l} launches; FLOPs count the VALID squeezed frames only (the recomputed halo rows are not counted)",
                         "profile": "profiles/r03_v3_trainstep_summary.txt (rocprofv3 --kernel-trace --stats of the same command)"})
            pmc = os.path.join(ROOT, "profiles", pmc_name)
            if os.path.exists(pmc):
                with open(pmc) as f:
                    pj = json.load(f)
                if "kernels" in pj:                                   # tools/decoder_pmc.py: the decoder's four fused kernels
                    kj = pj["kernels"].get("wn_stack_fwd", {})
                    roof["traffic"] = kj.get("hbm_bytes_per_launch")
                    roof["mfma_busy_fraction"] = kj.get("mfma_busy_fraction_of_wave_lifetime")   # SQ_VALU_MFMA_BUSY_CYCLES / wave lifetime
                    roof["traffic_source"] = pj.get("source")
                else:
                    roof["traffic"] = pj.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = pj.get("source")
        roof["step"] = {"algorithmic_flops_per_step": flops_step, "achieved": flops_step / (wall / args.steps) / 1e12,
                        "frac": flops_step / (wall / args.steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                        "how": "SURVEY §8d: 3 x forward FLOPs over the valid frames / tokens of the rotated batches, / wall time per step"}
        line["roofline"] = roof
        m = mas_leg(dev, wl, rank)
        line["mas"] = {"metric": "mas_alignments_per_sec", "value": world * m["alignments_per_sec"], "unit": "alignments/s",
                       "ms_per_batch": m["ms_per_batch"], "cpu_baseline": m.get("cpu"),
                       "roofline": {"bound": "hbm", "achieved": m["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": m["hbm_frac"], "traffic": None, "kernel": "gt_mas_dp_kernel + gt_mas_expand_kernel",
                                    "algorithmic_bytes_per_launch": m["algorithmic_bytes"]}}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batches[0], model, wl)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
