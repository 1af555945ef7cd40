"""Dev tool: idle time inside one replayed step, from a rocprofv3 --kernel-trace database.
usage: timeline.py <dir> — takes a step in the middle of the run (between two gt_step_zero launches) and reports the span,
the time during which NO kernel was running, and the largest idle gaps with the kernels around them."""
import glob, sqlite3, sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
packs = [i for i, r in enumerate(rows) if "gt_step_zero" in r[0]]         # one per step, at its head (round 3; before: the packing launch)
if len(packs) < 3:
    packs = [i for i, r in enumerate(rows) if "gt_pack_conv_weights_multi" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else len(packs) // 2
i0, i1 = packs[k], packs[k + 1]                     # default: a step in the middle of the run (the replayed ones)
step = rows[i0:i1]
t0, t1 = step[0][1], max(r[2] for r in step)
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, kernel time {sum(r[2] - r[1] for r in step) / 1e6:.3f} ms, "
      f"queues {sorted({r[3] for r in step})}, streams {sorted({r[4] for r in step})}")
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
gaps = []
prev = None
for r in sorted(step, key=lambda r: r[1]):
    s, e = r[1], r[2]
    if cur_e is None:
        cur_s, cur_e, last = s, e, r
    elif s <= cur_e:
        if e > cur_e:
            cur_e, last = e, r
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last[0][:60], r[0][:60]))
        cur_s, cur_e, last = s, e, r
busy += cur_e - cur_s
print(f"busy (any kernel running) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps "
      f"(mean {sum(g[0] for g in gaps) / max(1, len(gaps)) / 1e3:.2f} us)")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0] / 1e3:7.1f} us  after {g[1]}  before {g[2]}")
# overlap: time with >= 2 kernels running
ev = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
n, last_t, over = 0, None, 0
for t, d in ev:
    if n >= 2:
        over += t - last_t
    n += d; last_t = t
print(f"time with >= 2 kernels running: {over / 1e6:.3f} ms")

# per-stream chains: time from the end of one kernel to the start of the next ON THE SAME STREAM (launch / dependency latency)
by_stream = {}
for r in sorted(step, key=lambda r: r[1]):
    by_stream.setdefault(r[4], []).append(r)
for sid, lst in sorted(by_stream.items(), key=lambda kv: -len(kv[1])):
    kt = sum(r[2] - r[1] for r in lst)
    gaps_s = [max(0, b[1] - a[2]) for a, b in zip(lst, lst[1:])]
    print(f"stream {sid}: {len(lst)} kernels, kernel time {kt / 1e6:.3f} ms, span {(lst[-1][2] - lst[0][1]) / 1e6:.3f} ms, "
          f"sum of gaps {sum(gaps_s) / 1e6:.3f} ms (median gap {sorted(gaps_s)[len(gaps_s) // 2] / 1e3 if gaps_s else 0:.2f} us)")
    agg = {}
    for r in lst:
        a = agg.setdefault(r[0][:70], [0, 0]); a[0] += 1; a[1] += r[2] - r[1]
    for name, (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"      {tot / 1e6:7.3f} ms {cnt:4d} x {tot / cnt / 1e3:7.1f} us  {name}")

if "--list" in sys.argv:                               # chronological listing of the step: start (us from the step's start), duration, name
    for r in sorted(step, key=lambda r: r[1]):
        nm = r[0].replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f}  q{r[3]}  {nm[:90]}")
