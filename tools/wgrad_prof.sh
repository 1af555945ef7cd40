#!/bin/bash
# dev: per-kernel durations of tools/wgrad_bench.py (the host-side planning of an eager flush is longer than its kernels)
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/wgp -- python3 $ROOT/tools/wgrad_bench.py "$@" > $OUT/wgrad_prof.log 2>&1) || { tail -5 $OUT/wgrad_prof.log; exit 1; }
python - <<PY
import glob, sqlite3
db = glob.glob("$OUT/wgp/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
import collections
seq = [(n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0], (e - s) / 1e3) for n, s, e in rows if "wgrad" in n or "weightnorm" in n]
# three configurations x 12 flushes each: print the last flush of each configuration
out, cur = [], []
for name, us in seq:
    cur.append((name, us))
    if "weightnorm" in name:
        out.append(cur); cur = []
for i in (11, 23, 35):
    if i < len(out):
        print(" | ".join(f"{n[-28:]} {us:7.1f}" for n, us in out[i]), " total", round(sum(us for _, us in out[i]), 1))
PY
rm -rf $OUT/wgp
