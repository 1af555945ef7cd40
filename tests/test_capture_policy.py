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
