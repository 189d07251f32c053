run() { python bench.py --no-cpu-baseline $@ 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$*]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', d['config']['episode_latency_ms'])"; }
for i in 1 2 3; do
for f in 2 3 4 6; do run --steps 20 --warmup 5 --streams 2 --inflight $f; done
done
for f in 2 3 4 6; do run --steps 300 --warmup 10 --streams 2 --inflight $f; done
