#!/bin/bash
# Round-6 evidence for the headline command (run on the GPU box; writes under gpurun_out/):
#  1. DEFAULT (overlapped) mode: rocprofv3 kernel trace + stats -> tools/step_cadence.py reproduces roofline.frac from the trace
#  2. serial mode: per-launch durations (kernel stats)
#  3. FETCH_SIZE / WRITE_SIZE passes of the serial mode (separate runs) -> HBM traffic per launch of the step kernels (tools/pmc_summary.py)
#  4. the same two counter passes + kernel stats for the ComplexF64 full-Arnoldi problem (tools/one_complex.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CMD="python bench.py --no-cpu-baseline --no-secondary --no-serial-pass --steps 8 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_default -o t -- $CMD > gpurun_out/r6_default.log 2>&1
python tools/step_cadence.py gpurun_out/r6_default --skip 3 > gpurun_out/r06_step_cadence.txt 2>&1
grep "^{" gpurun_out/r6_default.log | tail -1 > gpurun_out/r06_bench_under_rocprof_default.json
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6_serial -o t -- $CMD > gpurun_out/r6_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r6_pmc$i -o c -- $CMD > gpurun_out/r6_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/r6_pmc1/c_counter_collection.csv gpurun_out/r6_pmc2/c_counter_collection.csv gpurun_out/r06_pmc_traffic.json > gpurun_out/r06_pmc_traffic.txt 2>&1
# complex full Arnoldi
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6c_serial -o t -- python tools/one_complex.py 6 > gpurun_out/r6c_serial.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/r6c_pmc$i -o c -- python tools/one_complex.py 4 > gpurun_out/r6c_pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/r6c_pmc1/c_counter_collection.csv gpurun_out/r6c_pmc2/c_counter_collection.csv gpurun_out/r06_pmc_traffic_complex.json > gpurun_out/r06_pmc_traffic_complex.txt 2>&1
find gpurun_out -name "*.db" -delete
# keep the small summaries, drop the bulky raw files
for d in r6_default r6_serial r6c_serial; do
  f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r06_rocprof_${d#r6}_kernel_stats.csv
done
f=$(find gpurun_out/r6_default -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r06_rocprof_default_kernel_trace.csv
rm -rf gpurun_out/r6_pmc1 gpurun_out/r6_pmc2 gpurun_out/r6c_pmc1 gpurun_out/r6c_pmc2 gpurun_out/r6_default gpurun_out/r6_serial gpurun_out/r6c_serial
tail -12 gpurun_out/r06_step_cadence.txt; grep k_pipe gpurun_out/r06_pmc_traffic.txt; grep k_pipe gpurun_out/r06_pmc_traffic_complex.txt
# round 6 extras: kernel + memory-copy trace of kiops (C4 complex, real) with the final library, and the host-side call sequence
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r6_kiops_c4 -- python tools/run_c4.py 6 > gpurun_out/r6_kiops_c4.log 2>&1
python tools/timeline.py gpurun_out/r6_kiops_c4 40 > gpurun_out/r06_kiops_c4_timeline.txt 2>&1
EXPV_MI_CALL_TRACE=1 python tools/run_c4.py 3 2> gpurun_out/r06_c4_calltrace.txt > /dev/null
EXPV_MI_CALL_TRACE=1 python tools/kiops_trace.py 2> gpurun_out/r06_kreal_calltrace.txt > /dev/null
rm -rf gpurun_out/r6_kiops_c4
