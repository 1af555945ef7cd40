"""dev: per-launch time of the fused WaveNet-layer kernels, graph-replayed, with L2-hot weights (one layer's images over and
over) and with streaming weights (48 different layers round-robin, as in the step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glow_tts_amd import _lib
if len(sys.argv) > 2:                      # wn_layer_bench.py quick <lib.so>: an experiment build (tools: exp_variant2.sh)
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
from glow_tts_amd import modules, ops, flow_impl

dev = torch.device("cuda:0")
L = _lib.lib()
H, n = 192, 4
wns = [modules.WN(160, H, 5, 1, n, 0, 0.05).to(dev) for _ in range(12)]
for w in wns:
    modules.prepare_all(w)
g = torch.Generator().manual_seed(1234)
TY = int(os.environ.get("WN_BENCH_TY", "800"))      # longest utterance in mel frames (800: cfg 2; 400: cfg 5's row count)
t_y = (torch.randint(TY * 3 // 16, TY // 2 + 1, (32,), generator=g) * 2); t_y[0] = TY
lens = [int(v) // 2 for v in t_y]
rc = ops.RowsCtx(torch.tensor(lens, dtype=torch.int32, device=dev), TY // 2, lengths_host=lens, round_to=512)
R = rc.R
x = (torch.randn(R, H, device=dev) * rc.rowmask[:, None]).to(torch.bfloat16)
acts = torch.empty(R, n * H, dtype=torch.bfloat16, device=dev)
t = torch.empty(R, H, dtype=torch.bfloat16, device=dev); s = torch.empty_like(t); xn = torch.empty_like(t)
dpre = (torch.randn(R, 2 * H, device=dev) * rc.rowmask[:, None] * 0.1).to(torch.bfloat16)
dx = torch.empty_like(t); dpo = torch.empty(R, 2 * H, dtype=torch.bfloat16, device=dev)
st = lambda: _lib.current_stream(dev)


def fwd(wn, i, res=True):
    il, rs = wn.in_layers[i], wn.res_skip_layers[i]
    _lib.check(L.gt_wn_layer_fwd(_lib.ptr(x), H, _lib.ptr(il.pc.fwd), _lib.ptr(il.bias), None, 0, None, 0, rc.Tp, _lib.ptr(rc.rowmask),
                                 _lib.ptr(acts), n * H, _lib.ptr(t), _lib.ptr(s), H, _lib.ptr(rs.pc_res.fwd) if res else None,
                                 rs.bias.data_ptr() if res else None, _lib.ptr(xn), H, R, H, 5, 0.05, 7, None, None, 0, None, st()), "fwd")


def bwd(wn, i, s2=True):
    il, rs = wn.in_layers[i + 1], wn.res_skip_layers[i]
    _lib.check(L.gt_wn_layer_bwd(_lib.ptr(dpre), 2 * H, _lib.ptr(il.pc.dgrad), _lib.ptr(x), H, _lib.ptr(rc.rowmask), _lib.ptr(dx), H,
                                 _lib.ptr(rs.pc_res.dgrad) if s2 else None, _lib.ptr(xn), H, _lib.ptr(t), _lib.ptr(s), H, _lib.ptr(dpo), None, 2 * H,
                                 R, H, 5, 0.05, 7, None, None, 0, None, st()), "bwd")


def timeit(name, fn, launches=48, replays=10):
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        fn(0)
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph, stream=stream):
            for k in range(launches):
                fn(k)
    torch.cuda.synchronize()
    for _ in range(2):
        gph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / (replays * launches) * 1e3:7.1f} us / launch   (R = {R})", flush=True)


def stack(wn):
    flow_impl.wn_fwd(rc, wn, x, None, True, 7, layers_only=True)


def layers(wn):
    wn.set_stack(False, False)
    try:
        flow_impl.wn_fwd(rc, wn, x, None, True, 7, layers_only=True)
    finally:
        wn.set_stack(True, True)


from glow_tts_amd import wgrad
_saved = {}
dsk = (torch.randn(R, H, device=dev) * rc.rowmask[:, None] * 0.1).to(torch.bfloat16)
via_all = (torch.randn(R, n * H, device=dev) * rc.rowmask[:, None] * 0.1).to(torch.bfloat16)


def bwd_wn(wn, stack_on):
    if id(wn) not in _saved:
        _saved[id(wn)] = flow_impl.wn_fwd(rc, wn, x, None, True, 7, layers_only=True)[1]
    wn.set_stack(stack_on, stack_on)
    try:
        q = wgrad.WgradQueue(dev, site=wn)
        q.__enter__()
        flow_impl.wn_bwd(rc, wn, _saved[id(wn)], dsk, dacts_skip=via_all)
        wgrad._ACTIVE.pop(); q.items = []                      # data-gradient chain only: drop the recorded weight-gradient jobs
    finally:
        wn.set_stack(True, True)


for w in wns:
    bwd_wn(w, True)
timeit("WN backward chain: ONE stack launch", lambda k: bwd_wn(wns[k % 12], True), launches=24)
STACK_ONLY = len(sys.argv) > 1 and sys.argv[1] == 'stack'
if not STACK_ONLY:
    timeit("WN backward chain: gate_bwd + 4 layer launches", lambda k: bwd_wn(wns[k % 12], False), launches=24)
timeit("WN forward, 4 layers: ONE stack launch", lambda k: stack(wns[k % 12]), launches=24)
if not STACK_ONLY:
    timeit("WN forward, 4 layers: four layer launches", lambda k: layers(wns[k % 12]), launches=24)
if len(sys.argv) > 1 and sys.argv[1] == 'stack':          # the stack kernels only (tools/stack_variants.sh)
    if os.environ.get("WN_BENCH_COLD"):
        # the same launches behind a 640 MB fill each (evicts L2 and the 256 MB Infinity Cache): in the training step a WaveNet's weights
        # and saved-activation lines are cold; subtract the fill's own time
        big = torch.empty(160 * 1024 * 1024, dtype=torch.float32, device=dev)
        timeit("640 MB fill alone", lambda k: big.fill_(float(k)), launches=24)
        timeit("fill + WN forward (stack)", lambda k: (big.fill_(float(k)), stack(wns[k % 12])), launches=24)
        timeit("fill + WN backward (stack)", lambda k: (big.fill_(float(k)), bwd_wn(wns[k % 12], True)), launches=24)
        # ... with the WaveNet's weight images touched again behind the fill (they come back into the Infinity Cache / one XCD's L2): how much
        # of the cold launch's extra time is the weights' first touch, and how much the activations' (inputs from HBM, stores to cold lines)
        def touch(wn, attr):
            acc = None
            for il in wn.in_layers:
                acc = getattr(il.pc, attr).view(torch.int32).sum() if acc is None else acc + getattr(il.pc, attr).view(torch.int32).sum()
            for rs in wn.res_skip_layers[:-1]:
                acc = acc + getattr(rs.pc_res, attr).view(torch.int32).sum()
            return acc
        timeit("fill + weight touch alone", lambda k: (big.fill_(float(k)), touch(wns[k % 12], "fwd")), launches=24)
        timeit("fill + weight touch + WN forward (stack)", lambda k: (big.fill_(float(k)), touch(wns[k % 12], "fwd"), stack(wns[k % 12])), launches=24)
        timeit("fill + weight touch + WN backward (stack)", lambda k: (big.fill_(float(k)), touch(wns[k % 12], "dgrad"), bwd_wn(wns[k % 12], True)), launches=24)
    sys.exit(0)
QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'
for frac in ((1,) if QUICK else (1, 2, 4, 8)):          # fewer workgroups, same weights per workgroup: per-CU streaming limit or chip-level L2 limit?
    Rs = R // frac // 64 * 64
    def f(k, Rs=Rs):
        il = wns[(k // 3) % 12].in_layers[k % 3]
        _lib.check(L.gt_wn_layer_fwd(_lib.ptr(x), H, _lib.ptr(il.pc.fwd), _lib.ptr(il.bias), None, 0, None, 0, rc.Tp, _lib.ptr(rc.rowmask),
                                     _lib.ptr(acts), n * H, _lib.ptr(t), _lib.ptr(s), H, None, None, None, H, Rs, H, 5, 0.0, 7, None, None, 0, None, st()), "fwd")
    def b(k, Rs=Rs):
        il = wns[(k // 3) % 12].in_layers[k % 3]
        _lib.check(L.gt_wn_layer_bwd(_lib.ptr(dpre), 2 * H, _lib.ptr(il.pc.dgrad), None, 0, _lib.ptr(rc.rowmask), _lib.ptr(dx), H,
                                     None, None, 0, None, None, 0, None, None, 0, Rs, H, 5, 0.0, 7, None, None, 0, None, st()), "bwd")
    timeit(f"fwd stage 1 only, streaming, {Rs // 64} workgroups", f)
    timeit(f"bwd stage 1 only, streaming, {Rs // 64} workgroups", b)
if QUICK:
    timeit("fwd fused, streaming weights (36 layers)", lambda k: fwd(wns[(k // 3) % 12], k % 3))
    sys.exit(0)
timeit("fwd fused, L2-hot weights", lambda k: fwd(wns[0], 0))
timeit("fwd fused, streaming weights (36 layers)", lambda k: fwd(wns[(k // 3) % 12], k % 3))
timeit("fwd stage 1 only, hot", lambda k: fwd(wns[0], 0, False))
timeit("fwd stage 1 only, streaming", lambda k: fwd(wns[(k // 3) % 12], k % 3, False))
timeit("bwd fused, hot", lambda k: bwd(wns[0], 0))
timeit("bwd fused, streaming", lambda k: bwd(wns[(k // 3) % 12], k % 3))
timeit("bwd stage 1 only, streaming", lambda k: bwd(wns[(k // 3) % 12], k % 3, False))
for w in wns:
    w.set_fused(False); modules.prepare_all(w)
timeit("two-kernel path: gate conv, streaming", lambda k: ops.conv_rows(x, wns[(k // 3) % 12].in_layers[k % 3].pc, rc,
       bias=wns[(k // 3) % 12].in_layers[k % 3].bias, gate=True, out=acts[:, :H], gate_t=t, gate_s=s, drop_p=0.05, seed=7))
timeit("two-kernel path: gate conv, hot", lambda k: ops.conv_rows(x, wns[0].in_layers[0].pc, rc, bias=wns[0].in_layers[0].bias, gate=True,
       out=acts[:, :H], gate_t=t, gate_s=s, drop_p=0.05, seed=7))
timeit("two-kernel path: res 1x1, streaming", lambda k: ops.conv_rows(acts[:, :H], wns[(k // 3) % 12].res_skip_layers[k % 3].pc_res, rc,
       bias=wns[(k // 3) % 12].res_skip_layers[k % 3].bias[:H], addend=x, mask=True, out=xn))
timeit("two-kernel path: in_layer dgrad k5, streaming", lambda k: ops.conv_rows(dpre, wns[(k // 3) % 12].in_layers[k % 3].pc, rc, dgrad=True,
       addend=x, mask=True, out=dx))
