"""The trainer's graph-capture policy on a realistic stream of batch shapes (CPU: keys and bookkeeping only, nothing is captured).
A length-bucketed sampler (data.DistributedBucketSampler with the reference's boundaries, train_ms_emo_lang_pitch.py:101-109) over an
LJSpeech-shaped length distribution produces tens of distinct (padded T_x, padded T_y, rounded row counts) keys; each capture costs
~3 steps of work and ~2 GiB, so the policy must keep captures rare (the round-2 defaults re-captured on most steps)."""
import os
import random
import sys
from collections import OrderedDict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glow_tts_amd import data, train  # noqa: E402

BOUNDS = [32, 300, 400, 500, 600, 700, 800, 900, 1000]


def _stream(epochs=3, n=13000, batch=32):
    rnd = random.Random(0)
    ty = [max(40, min(1000, int(rnd.gauss(560, 180)))) for _ in range(n)]
    tx = [max(10, int(t / 5.8 + rnd.gauss(0, 6))) for t in ty]
    smp = data.DistributedBucketSampler(ty, batch, list(BOUNDS), 1, 0, True)
    for ep in range(epochs):
        smp.set_epoch(ep)
        for b in smp:
            yield [tx[i] for i in b], [ty[i] // 2 * 2 for i in b]


def _simulate(max_graphs=8, capture_after=1, **key_kw):
    pol, resident = train.CapturePolicy(capture_after), OrderedDict()
    captures = replays = eager = 0
    keys = set()
    for bx, by in _stream():
        key, Tx, Ty = train.graph_key((bx, by), (len(bx), max(bx)), (len(by), 80, max(by)), (), True, **key_kw)
        assert Tx >= max(bx) and Ty >= max(by) and Ty % 2 == 0
        keys.add(key)
        if key in resident:
            resident.move_to_end(key); replays += 1
        elif pol.admit(key):
            resident[key] = True; captures += 1; replays += 1
            while len(resident) > max_graphs:
                resident.popitem(last=False)
        else:
            eager += 1
    return dict(keys=len(keys), captures=captures, replays=replays, eager=eager, hit=replays / (replays + eager))


def test_sampler_capture_config_keeps_captures_rare():
    cfg = train.sampler_capture_config(BOUNDS)
    r = _simulate(max_graphs=cfg["max_graphs"], capture_after=cfg["capture_after"], ty_boundaries=cfg["ty_boundaries"], pad_tx=cfg["pad_tx"],
                  row_round=cfg["row_round"])
    steps = r["replays"] + r["eager"]
    assert steps > 1000 and r["keys"] <= 40, r
    assert r["captures"] <= 40 and r["hit"] >= 0.95, r              # a capture every ~40 steps at worst, 95 % of the steps replay


def test_first_sight_capture_with_eight_graphs_thrashes():
    """what the policy replaces (round 2's defaults): documented, so that the numbers in train.sampler_capture_config stay honest"""
    r = _simulate(max_graphs=8, capture_after=1, row_round=512, pad_tx=16, pad_ty=32)
    assert r["captures"] > 0.3 * (r["replays"] + r["eager"]), r


def test_capture_policy_admits_on_the_nth_sight_and_forgets():
    pol = train.CapturePolicy(3, max_tracked=4)
    assert [pol.admit("a") for _ in range(3)] == [False, False, True]
    assert pol.admit("a") is False                                    # admitted keys start over (the trainer only asks while not resident)
    for k in "bcdefg":
        pol.admit(k)
    assert len(pol.seen) <= 4


def test_flat_buffer_order_decoder_conditioning_layers_before_the_flush_parameters():
    """train.Trainer's flat buffer (host logic, no GPU): everything but the decoder first, then the decoder's conditioning layers
    (their gradients arrive through autograd at the end of the backward), then the parameters the decoder's batched weight-gradient
    flush completes — the tail [dec_cov, end) that the early optimizer pass and the end-of-step packing cover; dec0 still marks the
    start of the decoder's slice for the phased all-reduce."""
    import torch
    from glow_tts_amd import train
    for gin in (0, 32):
        cfg = dict(train.BASE_MODEL, n_blocks_dec=2, n_layers_enc=1, gin_channels=gin)
        m = train.build_model(cfg, device="cpu")
        tr = train.Trainer(m, graph=False)
        names = {id(p): n for n, p in m.named_parameters()}
        ns = [names[id(p)] for p in tr.buckets.params]
        assert sorted(ns) == sorted(names.values())                         # every parameter exactly once
        assert tr.buckets.n_accum <= tr.dec0 <= tr.dec_cov <= len(ns)
        assert not any(n.startswith("decoder.") for n in ns[tr.buckets.n_accum:tr.dec0])
        assert all(n.startswith("decoder.") and ".cond_layer." in n for n in ns[tr.dec0:tr.dec_cov])
        assert all(n.startswith("decoder.") and ".cond_layer." not in n and "cond_layer1" not in n for n in ns[tr.dec_cov:])
        assert (tr.dec_cov > tr.dec0) == (gin > 0)
        assert tr.dec0_off == tr.buckets.offsets[tr.dec0] and tr.dec_cov_off == tr.buckets.offsets[tr.dec_cov]
        # the parameters are views of the flat buffer in that order, and a torch in-place write into the covered tail is seen by the
        # version check that guards the end-of-step packing (a write through .data is not: Trainer.invalidate_packed)
        v0 = tr._param_version()
        with torch.no_grad():
            tr.buckets.params[tr.dec_cov].mul_(1.0)
        assert tr._param_version() != v0
        v1 = tr._param_version()
        with torch.no_grad():
            tr.buckets.params[0].mul_(1.0)                                    # not a parameter of the covered tail
        assert tr._param_version() == v1


def test_step_head_cpu_fallback_matches_the_separate_calls():
    """ops.step_head off the GPU: the separate fills + the seed bump it replaces (one launch, gt_step_zero, on the GPU — tested there)."""
    import torch
    from glow_tts_amd import ops
    d = torch.device("cpu")
    s0 = int(ops.seed_word(d).item())
    x = torch.ones(12)
    ops.step_head(d, extra=x)
    assert int(x.count_nonzero()) == 0 and (int(ops.seed_word(d).item()) - s0) & 0xffffffff == 0x632BE5AB
    a = ops.zeros_small((5,), torch.float32, d)                            # the arena is open: slices of it are handed out
    assert a.shape == (5,) and int(a.count_nonzero()) == 0
    ops.step_head(d, extra=None, bump=False)
    assert int(ops.seed_word(d).item()) & 0xffffffff == (s0 + 0x632BE5AB) & 0xffffffff
    ops.arena_end(d)


def test_flat_adamw_early_range_and_the_rest_cover_the_active_runs_once():
    """FlatAdamW.step_early + step() (host logic; the kernel launch is replaced by a recorder): the early range and what step() launches
    afterwards partition exactly the runs of parameters that have a gradient — nothing twice, nothing of a gradient-less parameter."""
    import torch
    from glow_tts_amd import train
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (70, 130, 64, 200, 90, 300)]
    gb = train.GradBuckets(ps, world=1)
    opt = train.FlatAdamW(gb, 1e-3, (0.9, 0.98), 1e-9)
    launched = []
    opt._launch = lambda s, e: launched.append((s, e))
    opt._begin = lambda: None
    gb.active = [True, False, True, True, True, True]                      # parameter 1 never gets a gradient
    lo, hi = gb.offsets[4], gb.total                                        # the tail: parameters 4 and 5
    opt.step_early(lo, hi)
    assert launched == [(lo, hi)]
    opt.step()
    covered = sorted(launched)
    want = [(gb.offsets[0], gb.offsets[1]), (gb.offsets[2], lo), (lo, hi)]
    assert covered == sorted(want), (covered, want)
    assert opt._early is None
    launched.clear()
    opt.step()                                                              # no early pass this time: the plain runs
    assert sorted(launched) == [(gb.offsets[0], gb.offsets[1]), (gb.offsets[2], gb.total)]
