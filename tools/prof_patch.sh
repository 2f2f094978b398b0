cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
EXPV_MI_PIPE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_serial -o t -- python tools/one_grid.py 1000 8 1 > gpurun_out/pp_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pp_live -o t -- python tools/one_grid.py 1000 8 1 > gpurun_out/pp_live.log 2>&1
find gpurun_out -name "*.db" -delete
head -30 gpurun_out/pp_serial/*/t_kernel_stats.csv | cut -c1-220
