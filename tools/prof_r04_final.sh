#!/bin/bash
# End-of-round-4 evidence for the headline command (run on the GPU box; writes under gpurun_out/):
#  1. DEFAULT (overlapped) mode: rocprofv3 kernel trace + stats -> tools/step_cadence.py reproduces roofline.frac from the trace
#  2. serial mode: per-launch durations (kernel stats)
#  3. FETCH_SIZE / WRITE_SIZE passes of the serial mode (separate runs) -> HBM traffic per launch of the step kernels (tools/pmc_summary.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 8 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_default -o t -- $CMD > gpurun_out/r4f_default.log 2>&1
python tools/step_cadence.py gpurun_out/r4f_default --skip 3 > gpurun_out/r04_step_cadence.txt 2>&1
grep "^{" gpurun_out/r4f_default.log | tail -1 > gpurun_out/r04_bench_under_rocprof_default.json
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_serial -o t -- $CMD > gpurun_out/r4f_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r4f_pmc$i -o c -- $CMD > gpurun_out/r4f_pmc$i.log 2>&1
done
find gpurun_out -name "*.db" -delete
python tools/pmc_summary.py gpurun_out/r4f_pmc1/c_counter_collection.csv gpurun_out/r4f_pmc2/c_counter_collection.csv gpurun_out/r04_pmc_traffic.json > gpurun_out/r04_pmc_traffic.txt 2>&1
tail -12 gpurun_out/r04_step_cadence.txt; grep k_pipe gpurun_out/r04_pmc_traffic.txt
