#!/bin/bash
# Sweep of the pipelining knobs at the reference's evaluation batch (fgn_test.py:49: batch = 4, 800x1328) and at cfg4's
# 8 episodes per step: graphs / eager x caller streams x episodes in flight, one box, alternating order.
# usage (GPU box): bash tools/micro/sweep_batch.sh <batch> <steps> [repeats]
set -uo pipefail
B=${1:-4}; STEPS=${2:-30}; REP=${3:-2}
for i in $(seq "$REP"); do
  for v in "--no-graphs --streams 1 --inflight 1" "--no-graphs --streams 2 --inflight 2" "--no-graphs --streams 2 --inflight 3" \
           "--graphs --streams 1 --inflight 1" "--graphs --streams 1 --inflight 2" "--graphs --streams 2 --inflight 2" "--graphs --streams 2 --inflight 3"; do
    python bench.py --workload cfg4 --batch "$B" --steps "$STEPS" --warmup 4 --no-cpu-baseline $v 2>/dev/null | grep '^{' | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('[B=$B $v]', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step  whole-step frac', d['roofline'].get('whole_step',{}).get('frac'), ' peak GiB', c.get('peak_memory_gib'))"
  done
done
