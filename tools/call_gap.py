import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# per expv: find combine kernels
out=[]
for i,(s,e,k) in enumerate(rows):
    if "k_combine1" in k:
        # previous pipe kernel end (max end among preceding 4 kernels named k_pipe_live)
        prev=[r for r in rows[max(0,i-6):i] if "k_pipe_live" in r[2]]
        nxt=[r for r in rows[i+1:i+12] if "k_pipe_live" in r[2]]
        if prev and nxt:
            pe=max(r[1] for r in prev)
            out.append(((s-pe)/1e3,(e-s)/1e3,(nxt[0][0]-e)/1e3,(nxt[0][0]-pe)/1e3))
import numpy as np
a=np.array(out[3:])
print("calls",len(a),"| last step end -> combine start %.1f us | combine duration %.1f us | combine end -> next first step start %.1f us | total between factorisations %.1f us"%tuple(a.mean(0)))
