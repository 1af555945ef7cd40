"""Dev tool: device memory held per captured step graph (one graph per rounded (text rows, mel rows) bucket)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import train

dev = torch.device("cuda:0")
model = train.build_model(device=dev).train()
tr = train.Trainer(model, graph=True)
print(f"model + optimizer: reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
for seed in range(int(os.environ.get("N", 5))):
    ids, t_x, y, t_y = train.synth_batch(32, 150, 800, seed, dev)
    lh = (t_x.tolist(), t_y.tolist())
    loss, _ = tr.step(ids, t_x, y, t_y, lengths_host=lh)
    torch.cuda.synchronize()
    print(f"batch {seed}: graphs {len(tr._captured)}  reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB  "
          f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB  loss {loss.item():.3f}", flush=True)
