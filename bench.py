#!/usr/bin/env python3
"""bench.py — Glow-TTS training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5dec]

One "step" = one full training step of the hot path (zero_grad, TextEncoder + FlowSpecDecoder
forward, logp + MAS, mle + duration loss, backward, gradient all-reduce, grad-norm, AdamW) on
one synthetic LJSpeech-shaped batch resident in HBM (seed 1234 + rank, random-init
weights of configs/base.json).  With N > 1 the driver launches this file under
torch.distributed.run, one rank per GPU: the batch is sharded by utterance (weak scaling,
B per GPU fixed) and the only collective is the RCCL gradient all-reduce.

Prints ONE JSON line on rank 0:
  value            = valid mel-frames / s over all ranks (BASELINE.json metric, first half)
  mas              = MAS alignments / s of gt_mas_f32 alone on the step's own lattice shape (second half)
  roofline         = dominant kernel (WaveNet in_layer implicit-GEMM conv, bf16 MFMA), timed live with
                     HIP events around each of its launches inside the timed steps
  cpu_baseline     = the oracle's training step (PyTorch-CPU fp32 restatement of the reference,
                     oracle/glowtts_ref.py + reference Cython MAS from oracle/_ref) on a bounded sample
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

WORKLOADS = {
    "cfg2": dict(B=32, T_x=150, T_y=800, desc="configs/base.json, LJSpeech-shaped synthetic batch, B=32/GPU, T_x<=150, T_y<=800, bf16 GEMMs"),
    "cfg3": dict(B=32, T_x=375, T_y=872, desc="configs/base_blank.json-shaped synthetic batch, B=32/GPU, T_x<=375, T_y<=872, bf16 GEMMs"),
    "cfg4": dict(B=20, T_x=235, T_y=500, gin=256,
                 desc="configs/base_blank_ms.json-shaped synthetic batch (multi-speaker, gin_channels=256, g ~ N(0,1) [B,256,1]), "
                      "B=20/GPU, T_x<=235, T_y<=500, bf16 GEMMs"),
    "cfg5dec": dict(B=32, T_x=127, T_y=400, gin=512, n_layers_enc=10, prosody=True, n_lang=10, lin=4,
                    desc="configs/base_blank_emo_lang_pitch.json-shaped synthetic batch, B=32/GPU, T_x<=127, T_y<=400, gin_channels=512, "
                         "10 encoder layers, language embedding (10 languages, 4 channels), 3 WaveNets per coupling block (wn + wn_energy + "
                         "wn_pitch) with g, l, pitch, energy inputs; WITHOUT the emotion embeddings and the stochastic duration / pitch / "
                         "energy predictors (SURVEY §8 f1)"),
}


def mas_leg(dev, wl, rank, iters=100):
    """MAS alone on a lattice of the workload's shape: alignments/s and achieved HBM GB/s."""
    from glow_tts_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(1234 + rank)
    B, T_x, T_y = wl["B"], wl["T_x"], wl["T_y"]
    t_x = torch.randint(max(1, int(T_x * 0.4)), T_x + 1, (B,), generator=g, dtype=torch.int32)
    t_y = torch.randint(max(1, T_y * 3 // 16), T_y // 2 + 1, (B,), generator=g, dtype=torch.int32) * 2
    t_y = torch.maximum(t_y, t_x + (t_x % 2))
    t_x[0], t_y[0] = T_x, T_y
    logp = (torch.randn(B, T_x, T_y, generator=g) * 5.0 - 100.0).to(dev)
    t_x, t_y = t_x.to(dev), t_y.to(dev)
    path = torch.empty_like(logp)
    ws_bytes = L.gt_mas_workspace_bytes(B, T_x, T_y)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)

    def run():
        rc = L.gt_mas_f32(_lib.ptr(logp), None, _lib.ptr(t_x), _lib.ptr(t_y), _lib.ptr(path), _lib.GT_DT_F32, None, None,
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
    return {"alignments_per_sec": B / (ms * 1e-3), "ms_per_batch": ms, "achieved_GBps": algo_bytes / (ms * 1e-3) / 1e9,
            "hbm_frac": algo_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo_bytes, "logp": logp, "t_x": t_x, "t_y": t_y}


def gate_conv_leg(dev, model, lh, T_y, p_drop, ragged, row_round, launches=20, replays=20):
    """The dominant kernel alone, on the step's own shapes and weights: the WN in_layer k=5 conv + gate of the first
    coupling block, `launches` back-to-back launches captured in ONE HIP graph (no host launch gaps, as in the step's own
    graph) and timed with HIP events on the stream they run on.  -> average launch duration in ms."""
    from glow_tts_amd import ops
    wn = model.decoder.flows[2].wn
    conv = wn.in_layers[0]
    lens = torch.tensor([v // 2 for v in lh[1]], dtype=torch.int32, device=dev)
    rc = ops.RowsCtx(lens, T_y // 2, lengths_host=[v // 2 for v in lh[1]], round_to=row_round) if ragged else ops.RowsCtx(lens, T_y // 2)
    H = wn.hidden_channels
    x = (torch.randn(rc.R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
    y = torch.empty(rc.R, H, dtype=torch.bfloat16, device=dev); t = torch.empty_like(y); s_ = torch.empty_like(y)

    def run():
        ops.conv_rows(x, conv.pc, rc, bias=conv.bias, gate=True, out=y, gate_t=t, gate_s=s_, drop_p=p_drop, seed=1)
    run(); torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        run()
        with torch.cuda.graph(g, stream=st):
            for _ in range(launches):
                run()
    torch.cuda.synchronize(dev)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / (replays * launches), rc.R


def cpu_baseline(ids, t_x, y, t_y, model, budget_utts=4, g=None, n_layers_enc=6, pitch=None, energy=None, lang=None):
    """The oracle's training step on the host cores: same weights, a bounded sample of the same batch."""
    from oracle import glowtts_ref as R
    from oracle import mas as omas
    kind = "port"
    mas_core = omas.ref_maximum_path_c if omas.ref_module() is not None else omas.oracle_maximum_path_c

    def mp(logp, mask):
        p = omas.oracle_maximum_path(logp.numpy(), mask.numpy(), core=mas_core)
        return torch.from_numpy(p).float()
    n = min(budget_utts, ids.shape[0])
    P = {k: v.detach().cpu().float().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    hp = dict(hidden_channels=192, n_layers_enc=n_layers_enc, n_heads=2, window_size=4, kernel_size=3, prenet=True, mean_only=True,
              n_blocks_dec=12, n_block_layers=4, kernel_size_dec=5, n_sqz=2)
    ids_c, tx_c, y_c, ty_c = ids[:n].cpu(), t_x[:n].cpu().long(), y[:n].cpu(), t_y[:n].cpu().long()
    Tx, Ty = int(tx_c.max()), int(ty_c.max())
    ids_c, y_c = ids_c[:, :Tx], y_c[:, :, :Ty]
    cores = torch.get_num_threads()
    times = []
    for it in range(2):
        t0 = time.perf_counter()
        out = R.train_forward(P, ids_c, tx_c, y_c, ty_c, mp, hp, g=None if g is None else g[:n].cpu(),
                              pitch=None if pitch is None else pitch[:n, :, :Ty].cpu(), energy=None if energy is None else energy[:n, :, :Ty].cpu(),
                              l=None if lang is None else torch.nn.functional.embedding(lang[:n].cpu(), P["emb_l.weight"]).unsqueeze(-1))
        out["loss"].backward()
        times.append(time.perf_counter() - t0)
        for v in P.values():
            v.grad = None
    dt = min(times)
    return {"value": float(ty_c.sum()) / dt, "unit": "mel-frames/s", "cores": cores, "kind": kind,
            "sample": f"fwd+loss+bwd (no optimizer) of the first {n} utterances of the batch ({int(ty_c.sum())} valid frames), "
                      f"PyTorch-CPU fp32 oracle + {'reference Cython' if mas_core is omas.ref_maximum_path_c else 'C port'} MAS, best of 2",
            "s_per_step": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying one HIP graph")
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

    from glow_tts_amd import _lib, ops, train
    _lib.lib()                                       # fails loudly if the HIP library is missing
    wl = WORKLOADS[args.workload]
    torch.manual_seed(1234)                          # identical initial weights on every rank
    gin = wl.get("gin", 0)
    cfg = dict(train.BASE_MODEL, gin_channels=gin, n_layers_enc=wl.get("n_layers_enc", 6), with_prosody_wn=bool(wl.get("prosody")),
               n_lang=wl.get("n_lang", 0), lin_channels=wl.get("lin", 0))
    model = train.build_model(cfg if (gin or wl.get("prosody")) else None, device=dev).train()
    if world > 1:
        for p in model.parameters():
            torch.distributed.broadcast(p.data, 0)
    use_graph = not args.no_graph
    tr = train.Trainer(model, world=world, graph=use_graph)
    ids, t_x, y, t_y = train.synth_batch(wl["B"], wl["T_x"], wl["T_y"], rank, dev)
    spk = torch.randn(wl["B"], gin, 1, generator=torch.Generator().manual_seed(4321 + rank)).to(dev) if gin else None
    cond = {"g": spk} if spk is not None else {}
    if wl.get("prosody"):                            # pitch ~ U[80,280) Hz with 30 % unvoiced zeros, energy ~ U[1,11)
        gp = torch.Generator().manual_seed(977 + rank)
        cond["pitch"] = ((80 + 200 * torch.rand(wl["B"], 1, wl["T_y"], generator=gp)) * (torch.rand(wl["B"], 1, wl["T_y"], generator=gp) > 0.3)).to(dev)
        cond["energy"] = (1 + 10 * torch.rand(wl["B"], 1, wl["T_y"], generator=gp)).to(dev)
        if wl.get("n_lang"):
            cond["l"] = torch.randint(0, 3, (wl["B"],), generator=gp).to(dev)
    lh = (t_x.tolist(), t_y.tolist())                # host copy of the lengths (a data loader has them): no per-step sync
    valid_frames = int(t_y.sum().item())
    padded_frames = wl["B"] * wl["T_y"]

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        tr.step(ids, t_x, y, t_y, lengths_host=lh, **cond)
    barrier()
    if not use_graph:
        ops.KERNEL_TIMER.enable("in_layer_gate_conv")  # HIP events around every launch of the dominant kernel
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, mle = tr.step(ids, t_x, y, t_y, lengths_host=lh, **cond)
    barrier()
    wall = time.perf_counter() - t0
    if use_graph:
        # a replayed HIP graph has no room for event pairs between its kernel nodes: the dominant kernel is
        # timed right after the timed region in 3 eager steps of the same trainer on the same batch
        # (same process, same stream, same data), with HIP events around each of its 48 launches per step.
        ops.KERNEL_TIMER.enable("in_layer_gate_conv")
        for _ in range(3):
            tr._step_impl(ids, t_x, y, t_y, lh, cond=cond)
    kt = ops.KERNEL_TIMER.collect()
    if world > 1:
        t = torch.tensor([wall, float(valid_frames)], device=dev, dtype=torch.float64)
        tmax = t.clone(); torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        tsum = t.clone(); torch.distributed.all_reduce(tsum, op=torch.distributed.ReduceOp.SUM)
        wall, total_valid = float(tmax[0]), float(tsum[1])
    else:
        total_valid = float(valid_frames)
    assert torch.isfinite(loss).item(), "training step diverged"

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        ragged = tr.cfg.ragged
        if ragged:                                    # utterances packed back to back: only their own frames (+ halos) are rows
            _, R_dec = ops.RowsCtx.row_starts([v // 2 for v in lh[1]], wl["T_y"] // 2, tr.cfg.row_round)
        else:
            R_dec = wl["B"] * (wl["T_y"] // 2 + 2 * ops.HALO)
        # in_layer 192 -> 384, k = 5: 368 640 MAC per squeezed frame.  Ragged layout: only the VALID squeezed
        # frames count as algorithmic work (halo / rounding rows the kernel also walks do not)
        rows_alg = valid_frames // 2 if ragged else R_dec
        flops_launch = 2.0 * rows_alg * 384 * 192 * 5
        line = {
            "metric": "mel_frames_per_sec_train_step",
            "value": total_valid / (wall / args.steps),
            "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": wl["desc"], "batch_per_gpu": wl["B"], "T_x": wl["T_x"], "T_y": wl["T_y"],
                       "valid_frames_per_gpu_step": valid_frames, "padded_frames_per_gpu_step": padded_frames,
                       "rows_layout": (f"ragged (decoder rows {R_dec})" if ragged else f"uniform (decoder rows {R_dec})"),
                       "parallelism": f"dp{world} (utterance-sharded, RCCL gradient all-reduce)",
                       "launch": (("one HIP graph per step" if world == 1 else "two HIP graphs per step (fwd+bwd | optimizer), RCCL all-reduce between them")
                                  if tr.graph_mode else "eager launches"),
                       "sub_graph": "upstream-equivalent live sub-graph of configs/base.json (one WN per coupling block, "
                                    "deterministic DurationPredictor) — SURVEY F1/F2/F4",
                       "final_loss": float(loss)},
            "padded_frames_per_sec": world * padded_frames / (wall / args.steps),
        }
        avg_ms, rows_launched = gate_conv_leg(dev, model, lh, wl["T_y"], model.decoder.flows[2].wn.p_dropout, ragged, tr.cfg.row_round)
        tf = flops_launch / (avg_ms * 1e-3) / 1e12
        traffic = None                               # HBM-side bytes per launch from the committed rocprofv3 --pmc passes
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_gate_conv_pmc.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        line["roofline"] = {"bound": "mfma", "achieved": tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": tf / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic,
                            "traffic_source": "profiles/r01_gate_conv_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                              "FETCH_SIZE x2 per the gfx950 correction; tools/gate_conv_pmc.py)",
                            "kernel": "gt_conv_gemm_kernel<64|128,128,true> (WN in_layer k=5 conv + gate, 48 launches/step; 64-row tiles when "
                                      "128-row tiles would give <= 256 workgroups, as on this workload)",
                            "algorithmic_flops_per_launch": flops_launch, "launch_ms": avg_ms,
                            "how": "20 launches on the step's shapes and weights captured in one HIP graph, HIP events around 20 replays",
                            "rows_per_launch": rows_launched,
                            "launch_ms_eager_in_step": (kt["ms"] / kt["count"]) if kt["count"] else None}
        m = mas_leg(dev, wl, rank)
        line["mas"] = {"metric": "mas_alignments_per_sec", "value": world * m["alignments_per_sec"], "unit": "alignments/s",
                       "ms_per_batch": m["ms_per_batch"],
                       "roofline": {"bound": "hbm", "achieved": m["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": m["hbm_frac"], "traffic": None, "kernel": "gt_mas_dp_kernel + gt_mas_expand_kernel",
                                    "algorithmic_bytes_per_launch": m["algorithmic_bytes"]}}
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ids, t_x, y, t_y, model, g=spk, n_layers_enc=wl.get("n_layers_enc", 6),
                                                pitch=cond.get("pitch"), energy=cond.get("energy"), lang=cond.get("l"))
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
