run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline $@ 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$*]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms')"; }
for i in 1 2 3; do
run --streams 2 --inflight 2
run --streams 3 --inflight 3
run --streams 2 --inflight 4
run --streams 4 --inflight 4
done
