#!/bin/bash
# caller streams x episodes in flight of bench.py on one box (long runs: the isolated instrumented first step holds
# every other caller stream back once, which weighs more on short runs with more streams)
run() { python bench.py --steps 300 --warmup 10 --no-cpu-baseline $@ 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$*]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['episode_latency_ms'])"; }
for i in 1 2; do
run --streams 2 --inflight 2
run --streams 3 --inflight 3
run --streams 3 --inflight 2
run --streams 2 --inflight 3
done
