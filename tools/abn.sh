#!/bin/bash
# N-way interleaved A/B of the headline on one box: tools/abn.sh REPS "<env 1>" "<env 2>" ...
R="$1"; shift
for i in $(seq $R); do
  for v in "$@"; do
    out=$(env $v python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 5 2>/dev/null)
    echo "$v | $(echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.0f matvecs/s  %.4f ms  expv_frac %.4f  serial_us %.2f' % (d['value'], d['ms_per_step'], r['expv_frac'], 1e3*r['serial']['avg_launch_ms']))")"
  done
done
