#!/bin/bash
# Instruction / wave-state counters of the long-window step kernels, fp64 against ComplexF64 (serial mode, n = 1e6, m = 30): is the complex
# step VALU-bound?  (SQ counters in separate passes; summary = per-launch means of the 24- and 32-column kernels)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/vc_$i -o c -- python tools/one_complex.py 3 > gpurun_out/vc_$i.log 2>&1
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/vd_$i -o c -- python tools/one_expv.py 1e6 3 > gpurun_out/vd_$i.log 2>&1
done
find gpurun_out -name "*.db" -delete
python - <<PY
import csv, glob, collections
for mode in ("vd","vc"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for f in sorted(glob.glob("gpurun_out/%s_*/c_counter_collection.csv"%mode)):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::","").replace("expv_mi::","")
            if "k_pipe" not in k or not (", 32, 2" in k or ", 24, 3" in k or ", 16, " in k): continue
            a=acc[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k in sorted(acc):
        print(mode, k[:40])
        for c,v in sorted(acc[k].items()): print("      %-32s %.4g" % (c, v[1]/v[0]))
PY
rm -rf gpurun_out/vc_* gpurun_out/vd_*
