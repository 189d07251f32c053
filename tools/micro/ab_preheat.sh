#!/bin/bash
# extra untimed steps in front of the warm-up (GPU clock state at the start of the timed region), driver-length runs
run() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; }
for i in 1 2 3; do run FGN_BENCH_PREHEAT=0; run FGN_BENCH_PREHEAT=60; done
