import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests/golden"); sys.path.insert(0, ROOT + "/tests")
import torch, torch.nn.functional as F
from fill import fill_module
from glow_tts_amd import attentions, ops, _lib
from glow_tts_amd.ops import conv_rows
from glow_tts_amd.flow_impl import conv_param_grads
from glow_tts_amd.modules import prepare_all
dev = torch.device("cuda:0")
ffn = fill_module(attentions.FFN(192, 192, 768, 3, p_dropout=0.1), "ffn.").to(dev)
prepare_all(ffn)
B, T = 3, 41
rc = ops.RowsCtx(torch.tensor([41, 17, 30], dtype=torch.int32, device=dev), T)
m = rc.rowmask2d[:, 2:2+T].unsqueeze(1)
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 192, T, generator=g).to(dev) * m).to(torch.bfloat16)
dy = (torch.randn(B, 192, T, generator=g).to(dev) * m).to(torch.bfloat16)
# torch reference on bf16-rounded operands
w1 = ffn.conv_1.weight.detach().to(torch.bfloat16).float().requires_grad_(True); b1 = ffn.conv_1.bias.detach().clone().requires_grad_(True)
w2 = ffn.conv_2.weight.detach().to(torch.bfloat16).float().requires_grad_(True); b2 = ffn.conv_2.bias.detach().clone().requires_grad_(True)
h = torch.relu(F.conv1d(x.float(), w1, b1, padding=1)) * m
h.retain_grad()
y = F.conv1d(h, w2, b2, padding=1) * m
y.backward(dy.float())
xr = rc.to_rows(x)
f1 = conv_rows(xr, ffn.conv_1.pc, rc, bias=ffn.conv_1.bias, relu=True, mask=True)
f2 = conv_rows(f1, ffn.conv_2.pc, rc, bias=ffn.conv_2.bias, mask=True)
def rel(a, b): return ((a - b).abs().max() / b.abs().max()).item()
print("f1", rel(rc.from_rows(f1).float(), h.detach()), "f2", rel(rc.from_rows(f2).float(), y.detach()))
dyr = rc.to_rows(dy)
df1 = conv_rows(dyr, ffn.conv_2.pc, rc, dgrad=True)
print("df1 (pre relu mask)", rel(rc.from_rows(df1).float() * m * (h > 0), h.grad * (h > 0)))
dc1 = torch.empty_like(df1)
L = _lib.lib()
L.gt_relu_drop_bwd(_lib.ptr(df1), df1.stride(0), _lib.ptr(f1), f1.stride(0), _lib.ptr(dc1), dc1.stride(0), rc.R, 768, 0.0, _lib.current_stream(dev))
ref_dc1 = h.grad * (h > 0)
got = rc.from_rows(dc1).float()
print("dc1", rel(got, ref_dc1))
err = (got - ref_dc1).abs().amax(dim=(0, 2))
print("per-channel max err (first 8 / worst 8):", err[:8].tolist(), err.topk(8))
gr = conv_param_grads(ffn.conv_1, xr, dc1, rc.R)
print("dw1", rel(gr[ffn.conv_1.weight], w1.grad), "db1", rel(gr[ffn.conv_1.bias], b1.grad))
e = (gr[ffn.conv_1.weight] - w1.grad).abs().amax(dim=(1, 2)); print("dw1 err by co top:", e.topk(8))
