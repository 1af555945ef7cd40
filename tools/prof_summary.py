"""Summarise a rocprofv3 --kernel-trace --stats run (csv or rocpd .db output): per-step time by kernel.
usage: prof_summary.py <dir> <steps> [rows] [--csv out.csv]   (dev tool)"""
import csv, glob, sqlite3, sys

d, nsteps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 25
out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None

rows = []
dbs = glob.glob(d + "/**/*.db", recursive=True)
f = [] if dbs else glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
else:
    db = dbs[0]
    c = sqlite3.connect(db)
    for name, calls, tot in c.execute("select name, count(*), sum(duration) from kernels group by name order by 3 desc"):
        rows.append((name, calls, float(tot), float(tot) / calls))
tot = sum(r[2] for r in rows)
print(f"total kernel time per step: {tot/nsteps/1e6:.3f} ms over {sum(r[1] for r in rows)/nsteps:.0f} launches/step")
for name, calls, t, avg in rows[:top]:
    print("%8.3f ms/step %6.0f calls/step avg %8.1f us  %s" % (t / nsteps / 1e6, calls / nsteps, avg / 1e3, name[:100]))
if out_csv:
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
        for name, calls, t, avg in rows:
            w.writerow([name, calls, int(t), round(avg, 1), round(100 * t / tot, 3)])
