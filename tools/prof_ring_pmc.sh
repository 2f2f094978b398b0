cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/rg_$i -o c -- python tools/one_grid.py 1000 3 1 > gpurun_out/rg_$i.log 2>&1
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/rd_$i -o c -- python tools/one_expv.py 1e6 3 > gpurun_out/rd_$i.log 2>&1
done
find gpurun_out -name "*.db" -delete
python - <<PY
import csv, glob, collections
for mode in ("rd","rg"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for f in sorted(glob.glob("gpurun_out/%s_*/c_counter_collection.csv"%mode)):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void expv_mi::dev::","")
            if "k_pipe" not in k or "32, 2" not in k: continue
            a=acc[k][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k in sorted(acc):
        print(mode, k[:34], "  ".join("%s %.3g"%(c.replace("SQ_",""), v[1]/v[0]) for c,v in sorted(acc[k].items())))
PY
