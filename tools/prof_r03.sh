#!/bin/bash
# Round-3 profiling passes (run on the GPU box; writes under gpurun_out/).
#  1. headline command: rocprofv3 kernel trace + stats, overlapped (default) and serial mode; FETCH_SIZE / WRITE_SIZE PMC passes
#     in serial mode (separate runs, as the MI355X guide prescribes)
#  2. general-sparse / irregular / Float32 operators (tools/general_sparse.py): kernel stats + FETCH / WRITE passes
#  3. dense GEMV (config 3, n = 32768): kernel stats + VALU-utilisation counters (why MFMA is not used)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_default -o t -- $CMD > gpurun_out/q_default.log 2>&1
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_serial -o t -- $CMD > gpurun_out/q_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/q_pmc$i -o c -- $CMD > gpurun_out/q_pmc$i.log 2>&1
done
GS="python tools/general_sparse.py rand5 band5_20000 band5_2000 powerlaw c2f32"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_gs -o t -- $GS > gpurun_out/q_gs.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/q_gs_pmc$i -o c -- $GS > gpurun_out/q_gs_pmc$i.log 2>&1
done
C3="python bench.py --config c3 --n3 32768 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_c3 -o t -- $C3 > gpurun_out/q_c3.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/q_c3_pmc -o c -- $C3 > gpurun_out/q_c3_pmc.log 2>&1
find gpurun_out -name "*.db" -delete
ls gpurun_out/q_*/ | head -60
