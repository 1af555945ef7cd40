"""Time the full train step eagerly (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import train

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
model = train.build_model(device=dev).train()
tr = train.Trainer(model, graph=bool(int(os.environ.get("GRAPH", "0"))), split_graph=bool(int(os.environ.get("SPLIT", "0"))))
ids, t_x, y, t_y = train.synth_batch(B, 150, 800, 0, dev)
lh = (t_x.tolist(), t_y.tolist())
for i in range(3):
    loss, mle = tr.step(ids, t_x, y, t_y, lengths_host=lh)
torch.cuda.synchronize()
print("loss", loss.item(), "mle", mle.item(), "gnorm", tr.grad_norm.item(), flush=True)
n = int(os.environ.get("N", 10))
t0 = time.perf_counter()
for i in range(n):
    loss, mle = tr.step(ids, t_x, y, t_y, lengths_host=lh)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"step {dt*1e3:.2f} ms  valid frames/s {t_y.sum().item()/dt:.0f}  loss {loss.item():.4f}", flush=True)
