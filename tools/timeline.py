"""Timeline of the last expv in a rocprofv3 --kernel-trace CSV: every kernel with start / end relative to the
first kernel of that expv, so launch gaps, overlap between the two pipeline streams and the host-side part
(between the last factorisation kernel and the combine) are visible."""
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda k: k.split("(")[0].replace("void expv_mi::dev::", "").replace("expv_mi::dev::", "")[:44]
comb = [i for i, r in enumerate(rows) if "k_combine" in r[2]]
if len(comb) < 2:
    sys.exit("need two expv calls in the trace")
a, b = comb[-2] + 1, comb[-1]
t0 = rows[a][0]
print("previous combine end -> first kernel start: %.1f us" % ((t0 - rows[comb[-2]][1]) / 1e3))
prev_end = t0
for s, e, k in rows[a:b + 1]:
    print("%-46s start %8.1f  end %8.1f  dur %6.1f" % (short(k), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
print("expv span first start -> combine end: %.1f us; period combine end -> combine end: %.1f us" %
      ((rows[b][1] - t0) / 1e3, (rows[b][1] - rows[comb[-2]][1]) / 1e3))
