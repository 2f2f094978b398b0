"""gpurun_out/parity_measured.jsonl (written by tests/_util.close during a `pytest -m gpu` run) -> profiles/rNN_parity_measured.txt:
every parity comparison of the GPU suite with its measured error and its bar, worst first."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = [json.loads(l) for l in open(src) if l.strip()]
seen = {}
for r in rows:                      # the last measurement of a comparison wins (re-runs append)
    seen[r["what"]] = r
rows = sorted(seen.values(), key=lambda r: -r["err"])
with open(dst, "w") as f:
    f.write("Measured parity errors of the -m gpu suite on MI355X (tests/_util.close), worst first.  %d comparisons;\n" % len(rows))
    f.write("%d above 1e-12, %d above 1e-13, %d above 1e-14.  Columns: measured error | bar | comparison.\n\n" % (
        sum(r["err"] > 1e-12 for r in rows), sum(r["err"] > 1e-13 for r in rows), sum(r["err"] > 1e-14 for r in rows)))
    for r in rows:
        f.write("%.3e  %.1e  %s\n" % (r["err"], r["tol"], r["what"]))
print(open(dst).read()[:2500])
