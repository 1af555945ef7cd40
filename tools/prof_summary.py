"""Summarise a rocprofv3 kernel_stats.csv: per-step time by kernel (dev tool)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
nsteps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot/nsteps/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)/nsteps:.0f} launches/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%8.3f ms/step %6.0f calls/step avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / nsteps / 1e6, int(r["Calls"]) / nsteps,
                                                              float(r["AverageNs"]) / 1e3, r["Name"][:100]))
