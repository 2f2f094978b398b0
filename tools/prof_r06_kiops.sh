#!/bin/bash
# round 6: kernel + memory-copy trace of kiops (C4 complex, real) -> profiles/r06_trace_kiops_call.txt
set -x
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_kiops; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/c4 -- python tools/run_c4.py 6 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/real -- python tools/kiops_trace.py > $O/real.log 2>&1
python tools/timeline.py $O/c4 70 > $O/c4_timeline.txt 2>&1
python tools/timeline.py $O/real 70 > $O/real_timeline.txt 2>&1
python tools/run_c4.py 20 > $O/c4_plain.log 2>&1
tail -2 $O/c4.log $O/real.log $O/c4_plain.log
