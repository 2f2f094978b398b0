#!/bin/bash
# round 6: kernel timeline of the two-kernel step on uniformly random columns, plain vs pipelined first kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06_two_kernel; rm -rf $O; mkdir -p $O
for opt in 0 1; do
  rocprofv3 --kernel-trace --output-format csv -d $O/p$opt -- python tools/one_random.py 3 random $opt > $O/p$opt.log 2>&1
  python tools/timeline.py $O/p$opt 62 > $O/timeline_p$opt.txt 2>&1
done
for kind in sprand; do :; done
