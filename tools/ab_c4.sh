#!/bin/bash
# interleaved A/B of BASELINE config 4 (kiops, complex, n = 1e6): tools/ab_c4.sh REPS "<env 1>" "<env 2>" ...
R="$1"; shift
for i in $(seq $R); do
  for v in "$@"; do
    echo "$v | $(env $v python tools/run_c4.py 2>/dev/null | tail -1 | cut -c1-160)"
  done
done
