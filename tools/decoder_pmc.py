"""Launch the decoder's four fused kernels (whole-WaveNet forward / backward, between-WaveNets forward / backward) on the cfg 2 step's
shapes for rocprofv3 --pmc passes (dev tool; counters in separate passes, the TCC block cannot hold FETCH_SIZE and WRITE_SIZE together):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out_f -- python3 tools/decoder_pmc.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out_w -- python3 tools/decoder_pmc.py
    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace -d out_s -- python3 tools/decoder_pmc.py
    python tools/decoder_pmc.py --parse out_f out_w out_s > profiles/r03_decoder_pmc.json

One 12-block FlowSpecDecoder forward + backward (B = 32, T_y <= 800, ragged rows rounded to 512: R = 9 728, dropout 0.05): 12 + 12
WaveNet launches, 13 + 13 boundary launches."""
import glob, json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

KERNELS = {"wn_stack_fwd": ("gt_wn_stack_fwd_kernel",), "wn_stack_bwd": ("gt_wn_stack_bwd_kernel",),      # (demangled | mangled names)
           "wn_boundary_fwd": ("gt_wn_boundary_fwd_kernel<true, true>", "gt_wn_boundary_fwd_kernelILb1ELb1E"),
           "wn_boundary_bwd": ("gt_wn_boundary_bwd_kernel<true, true>", "gt_wn_boundary_bwd_kernelILb1ELb1E")}


def counters(d):
    db = glob.glob(d + "/**/*.db", recursive=True)[0]
    c = sqlite3.connect(db)
    out = {}
    for key, pats in KERNELS.items():
        rows, dur = [], None
        for pat in pats:
            rows = c.execute("select counter_name, avg(counter_value), count(distinct dispatch_id) from pmc_events where name like ? "
                             "group by counter_name", ("%" + pat + "%",)).fetchall()
            dur = c.execute("select avg(end - start), count(*) from kernels where name like ?", ("%" + pat + "%",)).fetchone()
            if rows:
                break
        out[key] = {"counters": {n: v for n, v, _ in rows}, "launches": rows[0][2] if rows else 0,
                    "avg_us_under_the_profiler": None if not dur or dur[0] is None else dur[0] / 1e3}
    return out


if "--parse" in sys.argv:
    i = sys.argv.index("--parse")
    f, w, s = (counters(sys.argv[i + k]) for k in (1, 2, 3))
    R_valid = None
    res = {"workload": "cfg2-shaped ragged rows rounded to 512 (R = 9728), one 12-block decoder forward + backward, dropout 0.05",
           "source": "rocprofv3 --pmc, three passes of tools/decoder_pmc.py (FETCH_SIZE | WRITE_SIZE | SQ_*); FETCH_SIZE doubled (16-B-per-lane loads: "
                     "gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted; SQ_BUSY_CYCLES / "
                     "SQ_WAVE_CYCLES count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (same guide, cycle-constants table)",
           "kernels": {}}
    for key in KERNELS:
        fk, wk, sk = f[key]["counters"], w[key]["counters"], s[key]["counters"]
        e = {"launches": s[key]["launches"], "avg_us_under_the_profiler": s[key]["avg_us_under_the_profiler"],
             "FETCH_SIZE_KB": fk.get("FETCH_SIZE"), "WRITE_SIZE_KB": wk.get("WRITE_SIZE"), **{k: v for k, v in sk.items()}}
        if fk.get("FETCH_SIZE") is not None and wk.get("WRITE_SIZE") is not None:
            e["hbm_bytes_per_launch"] = (2.0 * fk["FETCH_SIZE"] + wk["WRITE_SIZE"]) * 1024.0
        waves, mfma_busy, wave_cyc = sk.get("SQ_WAVES"), sk.get("SQ_VALU_MFMA_BUSY_CYCLES"), sk.get("SQ_WAVE_CYCLES")
        if waves and mfma_busy and wave_cyc:
            # per wave: matrix-pipe busy cycles over the wave's lifetime (quad-cycles x 4): the share of its life a wave keeps its SIMD's
            # matrix pipe busy (one wave per SIMD in these kernels, so this IS the pipe's utilisation on the CUs that hold a workgroup)
            e["mfma_busy_cycles_per_wave"] = mfma_busy / waves
            e["wave_lifetime_cycles"] = 4.0 * wave_cyc / waves
            e["mfma_busy_fraction_of_wave_lifetime"] = mfma_busy / (4.0 * wave_cyc)
        res["kernels"][key] = e
    print(json.dumps(res, indent=1))
    sys.exit(0)

import torch
from glow_tts_amd import flow_impl, models, modules, ops, wgrad
dev = torch.device("cuda:0")
nb = 12
dec = models.FlowSpecDecoder(80, 192, 5, 1, nb, 4, p_dropout=0.05).to(dev).train()
for b in range(nb):
    torch.nn.init.normal_(dec.flows[3 * b + 2].end.weight, std=0.01)
modules.prepare_all(dec)
g = torch.Generator().manual_seed(1234)
t_y = (torch.randint(150, 401, (32,), generator=g) * 2); t_y[0] = 800
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), 400, lengths_host=lens, round_to=512)
rows = torch.randn(rc.R, 160, device=dev) * rc.rowmask[:, None]
for _ in range(2):
    ld = torch.zeros(rc.B, device=dev)
    z, blocks = flow_impl.decoder_fwd_fused(rc, dec, rows, [None] * nb, ld, True, 7)
    dld = torch.zeros(rc.B, device=dev)
    with wgrad.WgradQueue(dev, site=dec):
        flow_impl.decoder_bwd_fused(rc, dec, blocks, torch.randn_like(rows) * rc.rowmask[:, None], dld, False)
torch.cuda.synchronize()
print("rows", rc.R)
