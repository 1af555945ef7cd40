"""Launch the dominant kernel (fused WaveNet layer forward, csrc/wn_layer.hip) on the cfg2 step's shapes for rocprofv3 --pmc
passes (dev tool; one counter per pass, the TCC block cannot hold both):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -- python3 tools/wn_layer_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -- python3 tools/wn_layer_pmc.py
then  python tools/wn_layer_pmc.py --parse out_f out_w > profiles/r02_wn_layer_pmc.json"""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
KERNEL = "gt_wn_layer_fwd_kernel<true>"


def parse(d, counter):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    r = c.execute("select avg(counter_value), count(*) from pmc_events where name like '%gt_wn_layer_fwd_kernelILb1%' and counter_name = ?",
                  (counter,)).fetchone()
    if not r or not r[1]:
        r = c.execute("select avg(counter_value), count(*) from pmc_events where name like '%gt_wn_layer_fwd_kernel<true>%' and counter_name = ?",
                      (counter,)).fetchone()
    return (r[0] if r and r[1] else None), (r[1] if r else 0)


if "--parse" in sys.argv:
    i = sys.argv.index("--parse")
    f, fc = parse(sys.argv[i + 1], "FETCH_SIZE")
    w, wc = parse(sys.argv[i + 2], "WRITE_SIZE")
    out = {"kernel": KERNEL + " (k=5 conv 192->384 + gate + residual 1x1, 64-row tiles)", "FETCH_SIZE_KB_per_launch": f,
           "WRITE_SIZE_KB_per_launch": w, "launches": fc,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/wn_layer_pmc.py; FETCH_SIZE doubled (16-B-per-lane "
                     "loads: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted",
           "workload": "cfg2-shaped ragged rows rounded to 512 (R = 9728), 36 launches over the weights of 12 WaveNets (streaming), "
                       "dropout 0.05",
           "expected": "reads: x 3.7 MB + weights 0.81 MB once per XCD (8 x) = 10.2 MB; writes: T, S, acts, x_next 4 x 3.7 = 14.9 MB"}
    if f is not None and w is not None:
        out["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    print(json.dumps(out))
    sys.exit(0)

import torch
from glow_tts_amd import _lib, modules, ops
dev = torch.device("cuda:0")
L = _lib.lib()
H, n = 192, 4
wns = [modules.WN(160, H, 5, 1, n, 0, 0.05).to(dev) for _ in range(12)]
for w in wns:
    modules.prepare_all(w)
g = torch.Generator().manual_seed(1234)
t_y = (torch.randint(150, 401, (32,), generator=g) * 2); t_y[0] = 800
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 400, lengths_host=lens, round_to=512)
R = rc.R
x = (torch.randn(R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
acts = torch.empty(R, n * H, dtype=torch.bfloat16, device=dev)
t = torch.empty(R, H, dtype=torch.bfloat16, device=dev); s = torch.empty_like(t); xn = torch.empty_like(t)
st = _lib.current_stream(dev)
for k in range(36):
    wn = wns[(k // 3) % 12]
    il, rs = wn.in_layers[k % 3], wn.res_skip_layers[k % 3]
    _lib.check(L.gt_wn_layer_fwd(_lib.ptr(x), H, _lib.ptr(il.pc.fwd), _lib.ptr(il.bias), None, 0, None, 0, rc.Tp, _lib.ptr(rc.rowmask),
                                 _lib.ptr(acts), n * H, _lib.ptr(t), _lib.ptr(s), H, _lib.ptr(rs.pc_res.fwd), rs.bias.data_ptr(), _lib.ptr(xn), H,
                                 R, H, 5, 0.05, 7, None, None, 0, None, st), "fwd")
torch.cuda.synchronize()
print("rows", R)
