"""Timeline of the LAST `nlast` kernel launches of a rocprofv3 --kernel-trace CSV: name, duration, gap to the previous end."""
import csv
import glob
import sys

d, nlast = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", ""), r.get("VGPR_Count", ""),
                         r.get("LDS_Block_Size", ""), r.get("Scratch_Size", "")))
rows.sort()
short = lambda k: k.split("(")[0].replace("void expv_mi::dev::", "").replace("expv_mi::dev::", "")[:70]
seq = rows[-nlast:]
t00 = seq[0][0]
prev = None
for s, e, k, g, v, l, sc in seq:
    print("%9.1f us  %-70s %8.2f us  gap %7.2f  grid %s vgpr %s lds %s scr %s" % ((s - t00) / 1e3, short(k), (e - s) / 1e3,
                                                                              0.0 if prev is None else (s - prev) / 1e3, g, v, l, sc))
    prev = e
print("span %.1f us, busy %.1f us" % ((seq[-1][1] - seq[0][0]) / 1e3, sum(e - s for s, e, *_ in seq) / 1e3))
