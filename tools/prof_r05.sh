#!/bin/bash
# Round-5 evidence for the headline command (run on the GPU box; writes under gpurun_out/):
#  1. DEFAULT (overlapped) mode: rocprofv3 kernel trace + stats -> tools/step_cadence.py reproduces roofline.frac from the trace
#  2. serial mode: per-launch durations (kernel stats)
#  3. FETCH_SIZE / WRITE_SIZE passes of the serial mode (separate runs) -> HBM traffic per launch of the step kernels (tools/pmc_summary.py)
#  4. the same two counter passes + kernel stats for the ComplexF64 full-Arnoldi problem (tools/one_complex.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 8 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_default -o t -- $CMD > gpurun_out/r5_default.log 2>&1
python tools/step_cadence.py gpurun_out/r5_default --skip 3 > gpurun_out/r05_step_cadence.txt 2>&1
grep "^{" gpurun_out/r5_default.log | tail -1 > gpurun_out/r05_bench_under_rocprof_default.json
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5_serial -o t -- $CMD > gpurun_out/r5_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r5_pmc$i -o c -- $CMD > gpurun_out/r5_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/r5_pmc1/c_counter_collection.csv gpurun_out/r5_pmc2/c_counter_collection.csv gpurun_out/r05_pmc_traffic.json > gpurun_out/r05_pmc_traffic.txt 2>&1
# complex full Arnoldi
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5c_serial -o t -- python tools/one_complex.py 6 > gpurun_out/r5c_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r5c_pmc$i -o c -- python tools/one_complex.py 4 > gpurun_out/r5c_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/r5c_pmc1/c_counter_collection.csv gpurun_out/r5c_pmc2/c_counter_collection.csv gpurun_out/r05_pmc_traffic_complex.json > gpurun_out/r05_pmc_traffic_complex.txt 2>&1
find gpurun_out -name "*.db" -delete
# keep the small summaries, drop the bulky raw files
for d in r5_default r5_serial r5c_serial; do
  f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_rocprof_${d#r5}_kernel_stats.csv
done
f=$(find gpurun_out/r5_default -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r05_rocprof_default_kernel_trace.csv
rm -rf gpurun_out/r5_pmc1 gpurun_out/r5_pmc2 gpurun_out/r5c_pmc1 gpurun_out/r5c_pmc2 gpurun_out/r5_default gpurun_out/r5_serial gpurun_out/r5c_serial
tail -12 gpurun_out/r05_step_cadence.txt; grep k_pipe gpurun_out/r05_pmc_traffic.txt; grep k_pipe gpurun_out/r05_pmc_traffic_complex.txt
