#!/bin/bash
# Round-4 profiling passes (run on the GPU box; writes under gpurun_out/).
#  1. headline command in the DEFAULT (overlapped) mode: rocprofv3 kernel trace + stats -> tools/step_cadence.py reproduces
#     roofline.frac from the trace (end-to-end cadence of consecutive k_pipe_live launches)
#  2. the same in serial mode (per-launch durations)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 8 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_default -o t -- $CMD > gpurun_out/r4_default.log 2>&1
python tools/step_cadence.py gpurun_out/r4_default --skip 3 > gpurun_out/r04_step_cadence.txt 2>&1
tail -12 gpurun_out/r04_step_cadence.txt
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_serial -o t -- $CMD > gpurun_out/r4_serial.log 2>&1
find gpurun_out -name "*.db" -delete
ls gpurun_out/r4_default/*/ | head
#  3. unstructured operators (tools/general_sparse.py): kernel stats + FETCH / WRITE passes -- uniformly random columns, irregular
#     (Zipf) rows, a shuffled banded operator and a shuffled 2-D grid (both reordered at creation: reorder.h)
GS="python tools/general_sparse.py rand5 powerlaw shuf_c2 shuf_grid"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4_gs -o t -- $GS > gpurun_out/r4_gs.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r4_gs_pmc$i -o c -- $GS > gpurun_out/r4_gs_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/r4_gs_pmc1/c_counter_collection.csv gpurun_out/r4_gs_pmc2/c_counter_collection.csv gpurun_out/r04_pmc_traffic_general_sparse.json > gpurun_out/r04_pmc_traffic_general_sparse.txt 2>&1
find gpurun_out -name "*.db" -delete
cat gpurun_out/r04_pmc_traffic_general_sparse.txt
