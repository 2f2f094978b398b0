#!/bin/bash
# Counters of the single-pass step on the banded C2 operator: diagonal (DIA) form against SELL slots (EXPV_MI_NO_DIA=1), serial mode.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in dia sell; do
  [ $mode = sell ] && export EXPV_MI_NO_DIA=1 || unset EXPV_MI_NO_DIA
  i=0
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
    i=$((i+1))
    EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/svd_${mode}_$i -o c -- python tools/one_expv.py 1e6 3 > gpurun_out/svd_${mode}_$i.log 2>&1
  done
done
find gpurun_out -name "*.db" -delete
python - <<PY
import csv, glob, collections
for mode in ("dia","sell"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for f in sorted(glob.glob("gpurun_out/svd_%s_*/c_counter_collection.csv"%mode)):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::","")
            if "k_pipe<" not in k: continue
            a=acc[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k in sorted(acc):
        print(mode, k[:40], "  ".join("%s %.3g"%(c.replace("SQ_",""), v[1]/v[0]) for c,v in sorted(acc[k].items())))
PY
