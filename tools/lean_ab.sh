#!/bin/bash
# A/B of the lean tile loop (pipe.hip PIPE_LEAN): window cost per step, fp64 and ComplexF64, overlapped and serial launches, product library
# against libexpv_mi_nolean.so (tools/build_variant.py nolean -DPIPE_LEAN=0).  usage: bash tools/lean_ab.sh > gpurun_out/lean_ab.txt
for mode in "" serial; do
  for lib in "" nolean; do
    echo "==== library: ${lib:-product (lean)}   mode: ${mode:-overlapped}"
    if [ -n "$lib" ]; then export EXPV_MI_LIB=$PWD/exponentialutilities.jl_amd/libexpv_mi_$lib.so; else unset EXPV_MI_LIB; fi
    python tools/window_cost.py 1000000 $mode 2>&1 | grep -v amdgpu.ids
  done
done
