"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) into per-kernel HBM traffic per launch.  Units/corrections (MI355X_MICROARCH.md §HBM):
the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read, so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::", "")
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    rd = 2.0 * f * 1024.0 / max(nf, 1)      # gfx950 correction: x2 on the read side
    wr = w * 1024.0 / max(nw, 1)
    out[k] = {"launches": nf, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
    print("%-44s launches %4d  read %8.1f MB  write %8.1f MB  total %8.1f MB" % (k[:44], nf, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
json.dump(out, open(sys.argv[3], "w"), indent=1)
