#!/bin/bash
# Same-box A/B of bench.py variants (boxes differ by +-3 %: only alternating runs on ONE box discriminate a few percent).
# usage (GPU box): bash tools/ab.sh <repeats> "<bench args>" "ENV=a ..." "ENV=b ..." ...   (use "-" for no env)
set -uo pipefail
REP=$1; ARGS=$2; shift 2
for i in $(seq "$REP"); do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e=""
    env $e python bench.py $ARGS --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  frac', d['roofline']['frac'])"
  done
done
