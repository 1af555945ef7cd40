"""Quick MAS timing on the GPU (dev tool, not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import monotonic_align as ma

dev = torch.device("cuda:0")
for (B, T_x, T_y) in [(32, 150, 800), (32, 375, 872), (256, 150, 800), (1024, 150, 800)]:
    g = torch.Generator().manual_seed(1234)
    v = (torch.randn(B, T_x, T_y, generator=g) * 5 - 100).to(dev)
    t_x = torch.full((B,), T_x, dtype=torch.int32, device=dev)
    t_y = torch.full((B,), T_y, dtype=torch.int32, device=dev)
    for _ in range(5):
        r = ma.maximum_path_lengths(v, t_x, t_y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        r = ma.maximum_path_lengths(v, t_x, t_y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byts = 8.0 * B * T_x * T_y
    print(f"B={B} T_x={T_x} T_y={T_y}: {ms*1e3:.1f} us/batch  {B/ms*1e3:.0f} align/s  {byts/ms/1e6:.1f} GB/s", flush=True)
