"""Launch the dominant kernel (a whole WaveNet forward in one launch, csrc/wn_stack.hip) on the cfg2 step's shapes for rocprofv3
--pmc passes (dev tool; one counter per pass, the TCC block cannot hold both):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -- python3 tools/wn_stack_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -- python3 tools/wn_stack_pmc.py
then  python tools/wn_stack_pmc.py --parse out_f out_w > profiles/r02_wn_stack_pmc.json"""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(d, counter):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    r = c.execute("select avg(counter_value), count(distinct dispatch_id) from pmc_events where name like '%gt_wn_stack_fwd_kernel%' and counter_name = ?",
                  (counter,)).fetchone()
    return (r[0] if r and r[1] else None), (r[1] if r else 0)


if "--parse" in sys.argv:
    i = sys.argv.index("--parse")
    f, fc = parse(sys.argv[i + 1], "FETCH_SIZE")
    w, wc = parse(sys.argv[i + 2], "WRITE_SIZE")
    out = {"kernel": "gt_wn_stack_fwd_kernel (4 x (k=5 conv 192->384 + gate) + 3 x residual 1x1, 52 owned rows per workgroup)",
           "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": fc,
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/wn_stack_pmc.py; FETCH_SIZE doubled (16-B-per-lane "
                     "loads: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted",
           "workload": "cfg2-shaped ragged rows rounded to 512 (R = 9728), 12 launches over the weights of 12 WaveNets, dropout 0.05",
           "expected": "reads: x0 with the 68/52 halo 4.9 MB + 4 layers of weights (3.2 MB) once per XCD = 30 MB; "
                       "writes: T, S, acts per layer 3 x 4 x 3.7 + x_1..3 3 x 3.7 = 56 MB"}
    if f is not None and w is not None:
        out["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    print(json.dumps(out))
    sys.exit(0)

import torch
from glow_tts_amd import flow_impl, modules, ops
dev = torch.device("cuda:0")
H, n = 192, 4
wns = [modules.WN(160, H, 5, 1, n, 0, 0.05).to(dev) for _ in range(12)]
for w in wns:
    modules.prepare_all(w)
g = torch.Generator().manual_seed(1234)
t_y = (torch.randint(150, 401, (32,), generator=g) * 2); t_y[0] = 800
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 400, lengths_host=lens, round_to=512)
x = (torch.randn(rc.R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
for w in wns:
    flow_impl.wn_fwd(rc, w, x, None, True, 7, layers_only=True)
torch.cuda.synchronize()
print("rows", rc.R)
