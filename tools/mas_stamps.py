"""Diagnostic build of the MAS kernel with phase cycle stamps (dev tool)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
so = "/tmp/libmas_stamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++20", "-shared", "-fPIC",
                       "-DMAS_STAMPS", *sys.argv[1:], "-I" + ROOT + "/include", ROOT + "/glow-tts_amd/csrc/mas.hip", "-o", so])
L = ctypes.CDLL(so)
dev = torch.device("cuda:0")
for (B, T_x, T_y) in [(32, 150, 800), (32, 375, 872)]:
    v = (torch.randn(B, T_x, T_y) * 5 - 100).to(dev)
    t_x = torch.full((B,), T_x, dtype=torch.int32, device=dev)
    t_y = torch.full((B,), T_y, dtype=torch.int32, device=dev)
    path = torch.empty_like(v)
    status = torch.zeros(16, dtype=torch.int32, device=dev)
    ws = torch.empty(B * (T_x + 1) * 4 + 256, dtype=torch.uint8, device=dev)
    for it in range(3):
        rc = L.gt_mas_f32(ctypes.c_void_p(v.data_ptr()), None, ctypes.c_void_p(t_x.data_ptr()), ctypes.c_void_p(t_y.data_ptr()),
                          ctypes.c_void_p(path.data_ptr()), 0, None, None, B, T_x, T_y, ctypes.c_int64(T_x * T_y), ctypes.c_int64(T_y),
                          ctypes.c_void_p(ws.data_ptr()), ctypes.c_size_t(ws.numel()), ctypes.c_void_p(status.data_ptr()), None)
        torch.cuda.synchronize()
    print(B, T_x, T_y, "rc", rc, "cycles fwd/backtrack/output:", status[1:9].tolist(), flush=True)
