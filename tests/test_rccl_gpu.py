"""The RCCL side of the data-parallel step on the one GPU of the test box: a world-size-1 `nccl` process group (backend "nccl" IS
RCCL on ROCm) under `Trainer(world=1, force_collectives=True)`, so that ReduceOp.AVG on the flat gradient buffer, the bf16 wire's
all-to-all + all-gather, the communication stream's hand-offs and "no collective is ever captured" (the all-reduce sits BETWEEN the
captured graphs) run against the real library — the reference's only collective is DDP's gradient all-reduce
(train_ms_emo_lang_pitch.py:72-74,166,309).  With one rank the mean is the identity: every schedule must reproduce the trainer that
issues no collective (to the run-to-run noise of the float atomics in fp32; to bf16 rounding of the gradients with the bf16 wire).  Also times GradBuckets.allreduce
of a cfg-2-sized (114 MB) and a cfg-5-sized (348 MB) buffer at world 1 = RCCL's launch + copy floor."""
import json
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        from glow_tts_amd import train
        cfg = dict(train.BASE_MODEL, n_blocks_dec=2, n_layers_enc=1, p_dropout=0.0, p_dropout_dec=0.0)

        def make():
            torch.manual_seed(0)
            m = train.build_model(cfg, device=dev)
            with torch.no_grad():
                for n, p in m.named_parameters():
                    if n.endswith("end.weight") or n.endswith("pre.proj.weight"):
                        p.normal_(0, 0.02)
            m.encoder.pre.p_dropout = 0.0
            return m

        stream = [train.synth_batch(4, *shape, sd, dev) for sd, shape in ((0, (40, 120)), (3, (40, 120)), (5, (37, 100)), (0, (40, 120)))]
        out = {}
        runs = (("plain", dict(graph=False)),                                              # no collective at all: the yardstick
                ("eager", dict(graph=False, force_collectives=True)),
                ("graph2", dict(graph=True, force_collectives=True)),                      # backward | all-reduce | optimizer
                ("graph3", dict(graph=True, split_graph=True, force_collectives=True)),    # phased: decoder slice travels beside the encoder's backward
                ("eager_bf16", dict(graph=False, force_collectives=True, grad_wire="bf16")),
                ("graph2_bf16", dict(graph=True, force_collectives=True, grad_wire="bf16")))
        info = {}
        for name, kw in runs:
            m = make()
            tr = train.Trainer(m, world=1, capture_after=1, **kw)
            tr.cfg.row_round = 32
            for b in stream:
                loss, _ = tr.step(*b, lengths_host=(b[1].tolist(), b[3].tolist()))
            torch.cuda.synchronize()
            assert tr.adam_steps == tr.n_steps == len(stream)
            if kw.get("graph"):
                n_graphs = len(next(iter(tr._captured.values()))[0])
                assert tr.graph_mode and n_graphs == (3 if kw.get("split_graph") else 2), n_graphs      # the collective is between graphs
                info[name] = n_graphs
            out[name] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        err = {k: (out[k] - out["plain"]).abs().max().item() for k in out if k != "plain"}

        # RCCL's floor for the step's exchange: the flat buffer of cfg 1-4 (28.6 M floats) and of cfg 5 (86.9 M) at world 1
        times = {}
        for label, n in (("cfg2_114MB", 28_600_000), ("cfg5_348MB", 86_900_000)):
            ps = [torch.nn.Parameter(torch.zeros(n // 4, device=dev)) for _ in range(4)]
            for wire in ("fp32", "bf16"):
                gb = train.GradBuckets(ps, 1, wire=wire, force_collectives=True)
                gb.flat.normal_()
                for _ in range(2):
                    gb.allreduce()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    gb.allreduce()
                torch.cuda.synchronize()
                times[f"{label}_{wire}_ms"] = (time.perf_counter() - t0) / 5 * 1e3
                del gb
        q.put((err, info, times, None))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(({}, {}, {}, traceback.format_exc()))


def test_rccl_world1_every_schedule_and_wire(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(29900 + (os.getpid() % 2000), q))
    p.start()
    err, info, times, tb = q.get(timeout=900)
    p.join(timeout=60)
    assert tb is None, tb
    for k in ("eager", "graph2", "graph3"):
        assert err[k] < 5e-3, (k, err)                  # fp32 wire, one rank: AVG is the identity (graphs differ from eager by bf16 GEMM noise only)
    for k in ("eager_bf16", "graph2_bf16"):
        assert err[k] < 2e-2, (k, err)                  # gradients rounded to bf16 twice on the wire
    assert info == {"graph2": 2, "graph3": 3, "graph2_bf16": 2}, info
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "rccl_world1.json"), "w") as f:
            json.dump({"param_err_vs_no_collective": err, "graphs": info, "allreduce_world1": times}, f, indent=1)
    print("RCCL world-1:", err, times)
