"""Build the HIP library in-tree: glow-tts_amd/libglowtts_hip.so (gfx950 only).

    python glow-tts_amd/build.py [--force] [--verbose]

hipcc cross-compiles without a GPU.  Objects are cached under glow-tts_amd/build/ and only
rebuilt when the source (or the public header) is newer.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libglowtts_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fno-gpu-rdc",
         "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function"]


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src, *extra))


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources under " + CSRC)
    hdrs = glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(CSRC, "*.hpp"))
    os.makedirs(OBJ, exist_ok=True)
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o, hdrs):
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[build] FAILED {s}\n{out.decode()}\n")
        elif verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    if procs or force or not os.path.exists(OUT):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
