"""End-to-end cadence of the overlapped single-pass step from a rocprofv3 --kernel-trace CSV.

In the product (overlapped) mode consecutive step kernels run on two streams: the workgroups of step j+1 become resident while
step j drains and wait on its flag, so a kernel's own duration includes waiting and says nothing about the step rate.  What a
step costs is the distance between the ENDS of consecutive step kernels of one factorisation,

    cadence(j) = End(step j) - End(step j-1),

and a factorisation of m steps spans End(step m) - Start(step 1).  This script groups the `k_pipe_live` launches of a trace into
factorisations (a gap of more than `--gap-us` between a kernel's start and the previous end starts a new one), prints the
cadence per step index and the average over the steps of all complete factorisations, and the roofline fraction that follows
from SURVEY 8d's algorithmic bytes per step (A_B + 8 n (j + 2), averaged over j = 1..m).  It is the rocprofv3 counterpart of the
HIP-event span / steps that bench.py reports as roofline.avg_launch_ms.

usage: python tools/step_cadence.py <rocprof output dir> [--m 30] [--n 1000000] [--nnz 4999994] [--kernel k_pipe_live] [--skip 2]
"""
import argparse
import csv
import glob
import sys

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--m", type=int, default=30)
ap.add_argument("--n", type=int, default=1_000_000)
ap.add_argument("--nnz", type=int, default=4_999_994)
ap.add_argument("--kernel", default="k_pipe_live")
ap.add_argument("--skip", type=int, default=2, help="factorisations to drop at the start (warm-up calls)")
ap.add_argument("--gap-us", type=float, default=60.0)
ap.add_argument("--peak", type=float, default=8000.0, help="GB/s")
args = ap.parse_args()

rows = []
for f in glob.glob(args.dir + "/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if args.kernel in r["Kernel_Name"] and "gate" not in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if not rows:
    sys.exit("no %s launches in %s" % (args.kernel, args.dir))
rows.sort(key=lambda r: r[1])                      # by END: overlapped launches start before their predecessor ends

facts, cur = [], [rows[0]]
for s, e in rows[1:]:
    if s - cur[-1][1] > args.gap_us * 1e3:         # this kernel started long after the previous one ended: a new factorisation
        facts.append(cur)
        cur = []
    cur.append((s, e))
facts.append(cur)
full = [f for f in facts if len(f) == args.m][args.skip:]
print("%d %s launches, %d factorisations, %d complete (m = %d) after dropping %d warm-up ones" %
      (len(rows), args.kernel, len(facts), len(full), args.m, args.skip))
if not full:
    sys.exit("no complete factorisation found (lengths: %s)" % sorted(set(len(f) for f in facts)))

s8 = 8
a_b = 12 * args.nnz + 4 * (args.n + 1)             # CSR32: 8 B value + 4 B column per entry + row pointers (bench.py: a_bytes)
bytes_step = [a_b + s8 * args.n * (j + 2) for j in range(1, args.m + 1)]
avg_bytes = sum(bytes_step) / args.m

print("\nstep   cadence us (mean over factorisations)   min      max    kernel's own duration us (mean)")
cad_all = []
for j in range(args.m):
    if j == 0:
        c = [(f[0][1] - f[0][0]) / 1e3 for f in full]          # the first step has no predecessor: its own duration
    else:
        c = [(f[j][1] - f[j - 1][1]) / 1e3 for f in full]
    d = [(f[j][1] - f[j][0]) / 1e3 for f in full]
    cad_all.append(sum(c) / len(c))
    print("%4d   %10.2f                              %7.2f  %7.2f   %10.2f" % (j + 1, sum(c) / len(c), min(c), max(c), sum(d) / len(d)))
span = [(f[-1][1] - f[0][0]) / 1e3 for f in full]
mean_span = sum(span) / len(span)
per_step = mean_span / args.m
own = sum((e - s) for f in full for s, e in f) / 1e3 / (len(full) * args.m)
print("\nfactorisation span (Start(step 1) .. End(step m)): mean %.1f us, min %.1f, max %.1f over %d factorisations" %
      (mean_span, min(span), max(span), len(full)))
print("end-to-end cadence = span / m = %.2f us per step   (mean of the kernels' own durations, which include waiting: %.2f us)" % (per_step, own))
print("algorithmic bytes per step (A_B + 8n(j+2), mean over j = 1..%d) = %.1f MB" % (args.m, avg_bytes / 1e6))
gbps = avg_bytes / (per_step * 1e-6) / 1e9
print("=> %.0f GB/s = %.3f of the %.0f GB/s roofline  [roofline.frac of bench.py, recomputed from this trace]" % (gbps, gbps / args.peak, args.peak))
