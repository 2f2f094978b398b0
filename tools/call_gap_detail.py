"""Kernel sequence around the end of one factorisation and the start of the next (rocprofv3 --kernel-trace CSV)."""
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::","")[:50], r.get("Queue_Id","?")))
rows.sort()
idx=[i for i,r in enumerate(rows) if "k_combine1" in r[2]]
i=idx[len(idx)//2]
t0=rows[i-4][0]
for s,e,k,q in rows[i-4:i+10]:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,q,k))
