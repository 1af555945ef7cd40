import sys; sys.path.insert(0, "/root/repo")
import torch
from glow_tts_amd import train
dev = torch.device("cuda:0")
model = train.build_model(None, device=dev).train()
tr = train.Trainer(model, world=1, graph=False)
names = {id(p): n for n, p in model.named_parameters()}
orig = tr.buckets.gather
def gather(lo=0, hi=None):
    b = tr.buckets
    hi2 = len(b.params) if hi is None else hi
    for i in range(lo, hi2):
        p = b.params[i]
        if p.grad is not None and p.grad.data_ptr() != b.view(i).data_ptr():
            print("stray grad:", names[id(p)], tuple(p.shape))
    return orig(lo, hi)
tr.buckets.gather = gather
ids, t_x, y, t_y = train.synth_batch(8, 60, 200, 0, dev)
tr.step(ids, t_x, y, t_y, lengths_host=(t_x.tolist(), t_y.tolist()))
torch.cuda.synchronize()
