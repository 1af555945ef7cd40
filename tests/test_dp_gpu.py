"""GPU test of the data-parallel TRAINER with world_size 2: two processes on the one MI355X of the test box, gloo as
the process group (RCCL refuses two ranks on one device; the collective library is not what is under test), so that
the phased backward, the three captured graphs and the all-reduces of the flat gradient buffer BETWEEN them run for
real.  Both ranks must end with identical parameters, the captured and the eager trainer must agree, and the result
must differ from training on one rank's data alone."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from glow_tts_amd import train
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
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

        batch = train.synth_batch(4, 40, 120, rank, dev)          # a different batch per rank
        lh = (batch[1].tolist(), batch[3].tolist())
        out = {}
        for name, graph, w in (("eager", False, world), ("graph", True, world), ("solo", False, 1)):
            m = make()
            tr = train.Trainer(m, world=w, graph=graph)
            n = 1 if graph else 4                                  # a graph trainer's first call = 3 warm-ups + 1 replay
            for _ in range(n):
                loss, _ = tr.step(*batch, lengths_host=lh)
            torch.cuda.synchronize()
            if graph:
                assert tr.graph_mode and len(next(iter(tr._captured.values()))[0]) == 3
            out[name] = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
        gathered = [torch.zeros_like(out["eager"]) for _ in range(world)]
        dist.all_gather(gathered, out["eager"])
        same_across_ranks = all(torch.equal(gathered[0], g) for g in gathered)
        graph_vs_eager = (out["graph"] - out["eager"]).abs().max().item()
        vs_solo = (out["solo"] - out["eager"]).abs().max().item()
        q.put((rank, same_across_ranks, graph_vs_eager, vs_solo, None))
        dist.destroy_process_group()
    except Exception as e:                                         # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, False, -1.0, -1.0, traceback.format_exc()))


def test_data_parallel_trainer_world2_on_one_gpu(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=300) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    for rank, same, gve, solo, err in res:
        assert err is None, err
        assert same, "ranks ended with different parameters"
        assert gve < 5e-3, gve                       # three graphs + collectives between them == eager phased step
        assert solo > 1e-4, "the gradient exchange changed nothing"
