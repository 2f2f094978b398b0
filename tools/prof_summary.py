"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the per-launch durations of the
last expv in the trace (so the j-dependence of each Krylov step is visible)."""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + d)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda k: k.split("(")[0].replace("void expv_mi::dev::", "").replace("expv_mi::dev::", "")[:60]
tot = defaultdict(lambda: [0, 0.0])
for s, e, k in rows:
    t = tot[short(k)]
    t[0] += 1
    t[1] += (e - s) / 1e3
print("%-62s %8s %12s %10s" % ("kernel", "calls", "total_us", "avg_us"))
for k, (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %8d %12.1f %10.2f" % (k, c, us, us / c))
# last expv: from the last k_sumsq to the end
starts = [i for i, r in enumerate(rows) if "k_sumsq" in r[2] or "k_fused_a2<double, false>" in r[2]
          or "k_fused_a2<expv_mi::cplx, false>" in r[2]]
idx = max(starts) if starts else 0
seq = rows[idx:]
print("\nlast expv: %d launches, span %.1f us, busy %.1f us" % (
    len(seq), (seq[-1][1] - seq[0][0]) / 1e3, sum(e - s for s, e, _ in seq) / 1e3))
for s, e, k in seq:
    print("  %-58s %8.2f us  gap_before %6.2f" % (short(k), (e - s) / 1e3, 0.0 if s == seq[0][0] else (s - prev) / 1e3)) if True else None
    prev = e
if len(sys.argv) > 2:
    json.dump({k: {"calls": c, "total_us": us, "avg_us": us / c} for k, (c, us) in tot.items()}, open(sys.argv[2], "w"), indent=1)
