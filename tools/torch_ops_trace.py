"""Dev tool: which host-side torch ops (copies, fills, cats ...) does one eager training step still launch, and from where?
(TorchDispatchMode + the innermost glow_tts_amd frame of each call.)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from glow_tts_amd import train

dev = torch.device("cuda:0")
model = train.build_model(device=dev).train()
tr = train.Trainer(model, graph=False)
ids, t_x, y, t_y = train.synth_batch(32, 150, 800, 0, dev)
lh = (t_x.tolist(), t_y.tolist())
for _ in range(3):
    tr.step(ids, t_x, y, t_y, lengths_host=lh)
torch.cuda.synchronize()
cnt = collections.Counter()
SKIP = ("view", "detach", "alias", "as_strided", "slice", "select", "reshape", "unsqueeze", "squeeze", "expand", "transpose", "permute",
        "empty", "_unsafe_view", "t.default", "is_", "stride", "size", "record_stream", "unbind", "split")


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in SKIP):
            site = "(torch/autograd)"
            for fr in reversed(traceback.extract_stack(limit=14)):
                if "glow-tts_amd" in fr.filename and "torch_ops_trace" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
                    break
            cnt[(name, site)] += 1
        return func(*args, **(kwargs or {}))


from glow_tts_amd import modules as _m
_orig_bwd = _m._RowsFn.backward


def _bwd(ctx, *grads):            # the runners' backward runs on the autograd engine's thread: enter the mode there too
    with Mode():
        return _orig_bwd(ctx, *grads)


_m._RowsFn.backward = staticmethod(_bwd)
with Mode():
    tr.step(ids, t_x, y, t_y, lengths_host=lh)
torch.cuda.synchronize()
tot = sum(cnt.values())
print("device-launching torch ops in one step:", tot)
for (name, site), c in cnt.most_common(int(os.environ.get('TOP', 70))):
    print(f"{c:4d}  {name:34s} {site}")
