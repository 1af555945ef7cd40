"""dev: build an experiment variant of the library: ONE source recompiled with extra -D flags, linked with the shipped objects.

    python tools/exp_variant.py NAME SOURCE[,SOURCE...] -DWNS_EXP=2 ...   ->  glow-tts_amd/build/exp/libglowtts_NAME.so

The bench tools take the library path as an argument (tools/wn_layer_bench.py quick <lib.so>)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glow-tts_amd"))
import build as B

name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build()
out_dir = os.path.join(B.OBJ, "exp")
os.makedirs(out_dir, exist_ok=True)
objs = []
paths = {}
for i, sname in enumerate(srcs):               # SOURCE=path/to/other.hip compiles that file in SOURCE's place (e.g. an older revision)
    if "=" in sname:
        srcs[i], paths[srcs[i]] = sname.split("=")[0], os.path.abspath(sname.split("=")[1])
for f in sorted(os.listdir(B.OBJ)):
    if f.endswith(".o") and f[:-2] not in srcs:
        objs.append(os.path.join(B.OBJ, f))
for s in srcs:
    o = os.path.join(out_dir, f"{s}_{name}.o")
    cmd = [B.HIPCC, *[f for f in B.FLAGS if not f.startswith("-Rpass")], *flags, "-I" + B.CSRC, "-c", paths.get(s, os.path.join(B.CSRC, s + ".hip")), "-o", o]
    subprocess.check_call(cmd)
    objs.append(o)
out = os.path.join(out_dir, f"libglowtts_{name}.so")
subprocess.check_call([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out, *objs])
print(out)
