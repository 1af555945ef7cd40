"""Launch the dominant kernel (WN in_layer k=5 conv + gate) on the cfg2 step's shapes for rocprofv3 --pmc passes (dev tool):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -- python tools/gate_conv_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -- python tools/gate_conv_pmc.py
then  python tools/gate_conv_pmc.py --parse out_f out_w > profiles/r01_gate_conv_pmc.json"""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(d, counter):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    r = c.execute("select avg(counter_value), count(*) from pmc_events where name like '%gt_conv_gemm_kernel%' and counter_name = ?",
                  (counter,)).fetchone()
    return (r[0] if r and r[1] else None), (r[1] if r else 0)


if "--parse" in sys.argv:
    i = sys.argv.index("--parse")
    f, fc = parse(sys.argv[i + 1], "FETCH_SIZE")
    w, wc = parse(sys.argv[i + 2], "WRITE_SIZE")
    out = {"kernel": "gt_conv_gemm_kernel<64,128,true> (the default tile at R = 9216: 216 workgroups of 128 rows <= 256 -> 64-row tiles)", "FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "launches": fc,
           "note": "rocprofv3 --pmc, one counter per pass; FETCH_SIZE is doubled for the 16-B-per-lane loads of this kernel "
                   "(gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted",
           "workload": "cfg2 batch (seed 1234), ragged rows rounded to 512: R = 9216, 192 -> 384 channels, k = 5, dropout 0.05"}
    if f is not None and w is not None:
        out["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
    print(json.dumps(out))
    sys.exit(0)

import torch
from glow_tts_amd import ops, train
from glow_tts_amd.modules import WNConvP
dev = torch.device("cuda:0")
ids, t_x, y, t_y = train.synth_batch(32, 150, 800, 0, dev)
lens_h = [v // 2 for v in t_y.tolist()]
lens = torch.tensor(lens_h, dtype=torch.int32, device=dev)
rc = ops.RowsCtx(lens, 400, lengths_host=lens_h, round_to=512)
H = 192
conv = WNConvP(H, 2 * H, 5, gate=True).to(dev); conv.prepare()
x = (torch.randn(rc.R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
yy = torch.empty(rc.R, H, dtype=torch.bfloat16, device=dev); t = torch.empty_like(yy); s = torch.empty_like(yy)
for _ in range(20):
    ops.conv_rows(x, conv.pc, rc, bias=conv.bias, gate=True, out=yy, gate_t=t, gate_s=s, drop_p=0.05, seed=1)
torch.cuda.synchronize()
print("rows", rc.R)
