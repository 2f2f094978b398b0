#!/bin/bash
# Round-2 profiling passes of the headline command (run on the GPU box; writes under gpurun_out/).
# kernel traces: default (overlapped) and serial mode; PMC: separate passes (FETCH_SIZE / WRITE_SIZE cannot share a pass,
# SQ counters 8 per pass), all in serial mode so that a kernel's counters are its own.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_default -o t -- $CMD > gpurun_out/p_default.log 2>&1
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_serial -o t -- $CMD > gpurun_out/p_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA MeanOccupancyPerCU OccupancyPercent LDSBankConflict"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/p_pmc$i -o c -- $CMD > gpurun_out/p_pmc$i.log 2>&1
done
find gpurun_out -name "*.db" -delete
ls gpurun_out/p_*/ | head -40
