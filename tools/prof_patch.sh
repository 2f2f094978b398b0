#!/bin/bash
# Patch form of the single-pass step on the bench's 2-D grid stencil (k = 1000, n = 1e6, m = 30): rocprofv3 kernel trace of the serial and
# of the overlapped mode (tools/step_cadence.py turns the second into the per-step cadence), and FETCH_SIZE / WRITE_SIZE passes of the
# serial mode (separate runs, as the MI355X guide prescribes) -> HBM traffic per launch of the k_pipe_ring kernels.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_serial -o t -- python tools/one_grid.py 1000 8 1 > gpurun_out/pp_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_live -o t -- python tools/one_grid.py 1000 8 1 > gpurun_out/pp_live.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  EXPV_MI_PIPE_SERIAL=1 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pp_pmc$i -o c -- python tools/one_grid.py 1000 4 1 > gpurun_out/pp_pmc$i.log 2>&1
done
find gpurun_out -name "*.db" -delete
{
  echo "Patch form, 2-D grid stencil k = 1000 (n = 1e6), m = 30: rocprofv3 passes of tools/one_grid.py (tools/prof_patch.sh)"
  echo
  echo "== overlapped mode: per-step cadence (tools/step_cadence.py) =="
  python tools/step_cadence.py gpurun_out/pp_live --skip 2
  echo
  echo "== serial mode: kernel stats =="
  head -12 gpurun_out/pp_serial/t_kernel_stats.csv | cut -c1-200
  echo
  echo "== serial mode: HBM traffic per launch (FETCH_SIZE x 2 read correction on gfx950, WRITE_SIZE as reported); contract per step, mean over j = 1..30: 204.0 MB =="
  python tools/pmc_summary.py gpurun_out/pp_pmc1/c_counter_collection.csv gpurun_out/pp_pmc2/c_counter_collection.csv gpurun_out/r04_pmc_traffic_patch_form.json
} > gpurun_out/r04_patch_form_profile.txt 2>&1
tail -40 gpurun_out/r04_patch_form_profile.txt
