"""dev: time of the step's weight packing launch (gt_pack_conv_weights_multi) on the cfg-2 model: pack_bench.py [lib.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from glow_tts_amd import train, modules

dev = torch.device("cuda:0")
model = train.build_model(None, device=dev).train()
model.prepare()
torch.cuda.synchronize()
plan = next(m._pack_plan for m in model.modules() if getattr(m, "_pack_plan", None) is not None)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 50
for _ in range(5):
    plan.run()
ev[0].record()
for _ in range(n):
    plan.run()
ev[1].record()
torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'default'}: pack launch(es) {ev[0].elapsed_time(ev[1]) / n * 1e3:.1f} us per call")
