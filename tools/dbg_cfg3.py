"""dev: the cfg-3 step run EAGERLY on the padded shapes a captured step sees (T_x 375 -> 384, T_y 872 -> 896), for GT_TRACE_CALLS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = train.build_model(None, device=dev).train()
tr = train.Trainer(model, world=1, graph=False)
for i in range(3):
    ids, t_x, y, t_y = train.synth_batch(32, 375, 872, 1000 * i, dev)
    lh = (t_x.tolist(), t_y.tolist())
    ids, y = tr._pad_time(ids, 384), tr._pad_time(y, 896)
    loss, mle = tr.step(ids, t_x, y, t_y, lengths_host=lh)
    torch.cuda.synchronize()
    print(i, float(loss), flush=True)
