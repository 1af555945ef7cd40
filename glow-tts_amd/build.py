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
         "-I" + os.path.join(ROOT, "include"), "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]      # per-kernel registers / scratch / LDS remarks, audited below


def resource_report(text):
    """[(kernel, {field: int})] from hipcc's kernel-resource-usage remarks."""
    import re
    out, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass-analysis", line)
        if not m:
            m = re.search(r"remark: +(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?): (.*?) \[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {}
            out.append((val, cur))
        elif cur is not None:
            try:
                cur[key] = int(val)
            except ValueError:
                cur[key] = val
    return out


def audit(reports, allow_scratch=()):
    """No shipped kernel may use scratch memory (private segment) or spill vector registers: the one GPU memory fault of
    round 1 came from a kernel built with 512 VGPRs + 42 spilled VGPRs + 508 spilled SGPRs + 172 B/lane of scratch
    (attention backward <8,2>); such a build now fails here instead of on the GPU."""
    bad = []
    for src, rep in reports.items():
        for name, r in rep:
            scratch = r.get("ScratchSize [bytes/lane]", 0)
            vsp = r.get("VGPRs Spill", r.get("VGPR Spill", 0))
            if (scratch or vsp) and not any(a in name for a in allow_scratch):
                bad.append((os.path.basename(src), name, scratch, vsp))
    return bad


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
    reports = {}
    for s, p in procs:
        out, _ = p.communicate()
        text = out.decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[build] FAILED {s}\n" + "\n".join(l for l in text.splitlines() if "-Rpass-analysis" not in l) + "\n")
            continue
        reports[s] = resource_report(text)
        with open(os.path.join(OBJ, os.path.basename(s)[:-4] + ".resources.txt"), "w") as f:
            for name, r in reports[s]:
                f.write(name + "  " + "  ".join(f"{k}={v}" for k, v in r.items()) + "\n")
        rest = [l for l in text.splitlines() if "-Rpass-analysis" not in l and l.strip()]
        if verbose and rest:
            print("\n".join(rest))
    if failed:
        raise RuntimeError("hipcc failed")
    # gt_flow_scalars: ONE thread's 4x4 Gauss-Jordan with pivoting in fp64 — a dynamically indexed private array (144 B), no spills
    bad = audit(reports, allow_scratch=("gt_flow_scalars_kernel", "gt_flow_scalars_multi_kernel"))
    if bad:
        for src, name, scratch, vsp in bad:
            sys.stderr.write(f"[build] {src}: kernel {name} uses scratch ({scratch} B/lane, {vsp} spilled VGPRs)\n")
        for src in {b[0] for b in bad}:                       # do not leave an object that would be linked next time
            o = os.path.join(OBJ, src[:-4] + ".o")
            if os.path.exists(o):
                os.remove(o)
        raise RuntimeError("kernels with scratch memory / register spills are not shipped (see DESIGN.md 4.5)")
    if procs or force or not os.path.exists(OUT):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", OUT, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
