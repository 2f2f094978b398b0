#!/bin/bash
# interleaved A/B of the grid-stencil (wave form) expv: tools/ab_stencil.sh REPS "<env 1>" "<env 2>" ...
R="$1"; shift
for i in $(seq $R); do
  for v in "$@"; do
    echo "$v | $(env $v python tools/bench_other.py c2stencil 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f matvecs/s  %.4f ms  frac %.4f' % (d['matvecs_per_s'], d['ms_per_expv'], d['frac_of_8TBps']))")"
  done
done
