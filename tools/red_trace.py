"""Phases of the grid reduction of the single-pass step, from the .red file tools/pipe_trace.py leaves next to its trace
(trace build).  Per step: last main-loop end -> last stage-1 ticket drawn -> its group sums stored -> last stage-2 ticket
drawn -> totals in LDS (= 'reduced')."""
import sys
import numpy as np
tr = np.loadtxt(sys.argv[1], dtype=np.int64)
rd = np.loadtxt(sys.argv[1] + ".red", dtype=np.int64)
print("step | main end (last wg) -> ticket1 drawn (last group) | -> group sums stored | -> ticket2 drawn (winner) | -> totals  || sum [us]")
tot = []
for q in sorted(set(rd[:, 0])):
    t = tr[tr[:, 0] == q]
    r = rd[rd[:, 0] == q]
    main_end = t[:, 3].max()
    w = r[r[:, 5] > 0]
    if not len(w):
        continue
    w = w[0]
    # the winner's own chain and the slowest group
    t1 = r[:, 2].max(); s1 = r[:, 3].max(); t2 = w[4]; fin = w[5]
    row = [(t1 - main_end) * 0.01, (s1 - t1) * 0.01, (t2 - s1) * 0.01, (fin - t2) * 0.01]
    tot.append(row + [sum(row)])
    print(q, " | ".join("%5.2f" % x for x in row), "|| %5.2f" % sum(row))
print("mean", " | ".join("%5.2f" % x for x in np.mean(np.array(tot)[1:], axis=0)))
