"""CPU tests of the checkpoint wire format (glow_tts_amd.checkpoint; reference utils.py:18-131): round trip through a
file, interchange with torch.optim.AdamW's state_dict in both directions, warm start with tensor growth."""
import os

import torch

SMALL = dict(hidden_channels=192, filter_channels=768, filter_channels_dp=256, kernel_size=3, p_dropout=0.1, n_blocks_dec=1,
             n_layers_enc=1, n_heads=2, p_dropout_dec=0.05, dilation_rate=1, kernel_size_dec=5, n_block_layers=4, n_sqz=2,
             prenet=True, mean_only=True, window_size=4, use_sdp=False)


def _trainer(seed):
    from glow_tts_amd import train
    torch.manual_seed(seed)
    m = train.build_model(SMALL, device="cpu")
    return m, train.Trainer(m, graph=False, total_steps=100)


def test_round_trip_and_torch_adamw_interchange(tmp_path):
    from glow_tts_amd import checkpoint
    m1, t1 = _trainer(1)
    g = torch.Generator().manual_seed(3)
    t1.opt.m.copy_(torch.randn(t1.opt.m.shape, generator=g))
    t1.opt.v.copy_(torch.rand(t1.opt.v.shape, generator=g))
    t1.opt.hyper[5] = 17.0
    t1.n_steps = 17
    path = os.path.join(tmp_path, "G_17.pth")
    checkpoint.save_checkpoint(t1, 2e-4, 17, path)

    ck = torch.load(path, map_location="cpu", weights_only=True)            # the reference's layout (utils.py:126-130)
    assert set(ck) == {"model", "iteration", "optimizer", "scheduler", "learning_rate"}
    assert list(ck["model"]) == list(m1.state_dict()) and "decoder.flows.2.wn.in_layers.0.weight_v" in ck["model"]

    # (a) a torch.optim.AdamW built the way the reference builds it accepts the optimizer state as it is
    ref_opt = torch.optim.AdamW(m1.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    ref_opt.load_state_dict(ck["optimizer"])
    names = [n for n, _ in m1.named_parameters()]
    pos = {id(p): i for i, p in enumerate(t1.buckets.params)}
    for k, p in enumerate(m1.parameters()):
        o = t1.buckets.offsets[pos[id(p)]]
        assert torch.equal(ref_opt.state[p]["exp_avg"].reshape(-1), t1.opt.m[o:o + p.numel()]), names[k]
        assert torch.equal(ref_opt.state[p]["exp_avg_sq"].reshape(-1), t1.opt.v[o:o + p.numel()]), names[k]
        assert float(ref_opt.state[p]["step"]) == 17.0

    # (b) a second trainer restores parameters, moments, step and schedule position from the file
    m2, t2 = _trainer(2)
    assert not torch.equal(m2.decoder.flows[2].start.weight_v, m1.decoder.flows[2].start.weight_v)
    lr, it = checkpoint.load_checkpoint(path, m2, t2)
    assert (lr, it) == (2e-4, 17) and t2.n_steps == 17
    for (n, a), b in zip(m1.named_parameters(), m2.parameters()):
        assert torch.equal(a, b), n
    i1, i2 = checkpoint._model_order(t1), checkpoint._model_order(t2)
    for a, b in zip(i1, i2):
        o1, o2, n = t1.buckets.offsets[a], t2.buckets.offsets[b], t1.buckets.params[a].numel()
        assert torch.equal(t1.opt.m[o1:o1 + n], t2.opt.m[o2:o2 + n]) and torch.equal(t1.opt.v[o1:o1 + n], t2.opt.v[o2:o2 + n])
    assert float(t2.opt.hyper[5]) == 17.0
    assert all(p.data_ptr() == t2.opt.flat_p[o:o + p.numel()].data_ptr() for p, o in zip(t2.buckets.params, t2.buckets.offsets)), \
        "loading must keep the parameters views of the flat buffer"

    # (c) a checkpoint written the reference's way (torch optimizer + scheduler state) loads into the trainer
    sch = torch.optim.lr_scheduler.OneCycleLR(ref_opt, max_lr=2e-4, total_steps=100)
    path2 = os.path.join(tmp_path, "G_ref.pth")
    torch.save({"model": m1.state_dict(), "iteration": 5, "optimizer": ref_opt.state_dict(), "scheduler": sch.state_dict(),
                "learning_rate": 1e-4}, path2)
    m3, t3 = _trainer(4)
    lr, it = checkpoint.load_checkpoint(path2, m3, t3)
    assert (lr, it) == (1e-4, 5)
    for a, b in zip(checkpoint._model_order(t1), checkpoint._model_order(t3)):
        o1, o3, n = t1.buckets.offsets[a], t3.buckets.offsets[b], t1.buckets.params[a].numel()
        assert torch.equal(t1.opt.m[o1:o1 + n], t3.opt.m[o3:o3 + n])


def test_warm_start_grows_tensors_and_skips_ignored_layers(tmp_path):
    """utils.warm_start_model + transfer_weight: a checkpoint with a smaller vocabulary warm-starts a bigger model (the
    new embedding rows are N(0,1)), `ignore_layers` keep their fresh values (configs/base_blank_ms.json "ignored_layer")."""
    from glow_tts_amd import checkpoint, train
    torch.manual_seed(0)
    small = train.build_model(SMALL, n_vocab=100, device="cpu")
    path = os.path.join(tmp_path, "pretrained.pth")
    torch.save({"model": small.state_dict(), "iteration": 1, "learning_rate": 1e-3}, path)
    big = train.build_model(SMALL, n_vocab=148, device="cpu")
    keep = big.encoder.proj_w.conv_1.weight.detach().clone()
    _, grown, skipped = checkpoint.warm_start_model(path, big, ignore_layers=["encoder.proj_w.conv_1.weight"],
                                                    generator=torch.Generator().manual_seed(5))
    assert grown == ["encoder.emb.weight"] and skipped == ["encoder.proj_w.conv_1.weight"]
    assert torch.equal(big.encoder.emb.weight[:100], small.encoder.emb.weight) and big.encoder.emb.weight.shape[0] == 148
    assert torch.equal(big.encoder.proj_w.conv_1.weight, keep)
    assert torch.equal(big.decoder.flows[2].wn.in_layers[1].weight_v, small.decoder.flows[2].wn.in_layers[1].weight_v)


def test_load_checkpoint_reports_keys_it_cannot_place(tmp_path):
    """A checkpoint with entries this model does not have (a cfg 5 file into a base-config model) and without some it has:
    load_checkpoint keeps the model's values for the missing ones, ignores the others and REPORTS both."""
    import warnings
    from glow_tts_amd import checkpoint
    m1, t1 = _trainer(2)
    path = os.path.join(tmp_path, "G_1.pth")
    checkpoint.save_checkpoint(t1, 2e-4, 1, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    dropped = "decoder.flows.0.logs"
    kept = ck["model"].pop(dropped)
    ck["model"]["proj_pitch.pre.weight"] = torch.zeros(3)
    torch.save(ck, path)
    m2, _ = _trainer(5)
    before = m2.state_dict()[dropped].clone()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        checkpoint.load_checkpoint(path, m2)
    rep = checkpoint.load_checkpoint.last_report
    assert rep == {"missing": [dropped], "ignored": ["proj_pitch.pre.weight"]}
    assert any("proj_pitch.pre.weight" in str(x.message) for x in w)
    assert torch.equal(m2.state_dict()[dropped], before)
    assert torch.equal(m2.state_dict()["decoder.flows.1.weight"], m1.state_dict()["decoder.flows.1.weight"])
