#!/bin/bash
# Where does a replayed hipGraph lose to eager launches on ONE stream?  rocprofv3 kernel stats of both modes at the same
# batch: per-kernel total durations side by side.  usage (GPU box): bash tools/micro/graph_vs_eager.sh <batch> <out dir>
set -uo pipefail
B=${1:-4}; OUT=${2:-gpurun_out/gve}; mkdir -p "$OUT"; ROOT=$(pwd)
for m in graphs no-graphs; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf "$ROOT/$OUT/$m" && rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/$m" --output-format csv -- python3 "$ROOT/bench.py" --workload cfg4 --batch "$B" --steps 12 --warmup 3 --no-cpu-baseline --$m --streams 1 --inflight 1 > "$ROOT/$OUT/$m.log" 2>&1 )
  cp "$OUT/$m"/*/*kernel_stats.csv "$OUT/${m}_kernel_stats.csv" 2>/dev/null
  grep '^{' "$OUT/$m.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms/step')"
  rm -rf "$OUT/$m"
done
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        d[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
    return d
g, e = load(f'{out}/graphs_kernel_stats.csv'), load(f'{out}/no-graphs_kernel_stats.csv')
print(f"{'kernel':70s} {'graph calls':>11s} {'graph ms':>9s} {'eager calls':>11s} {'eager ms':>9s} {'ratio':>6s}")
tg = te = 0
for k in sorted(set(g) | set(e), key=lambda k: -(g.get(k, (0, 0))[1] + e.get(k, (0, 0))[1])):
    a, b = g.get(k, (0, 0.0)), e.get(k, (0, 0.0))
    tg += a[1]; te += b[1]
    if a[1] + b[1] > 2.0:
        print(f'{k[:70]:70s} {a[0]:11d} {a[1]:9.2f} {b[0]:11d} {b[1]:9.2f} {a[1] / b[1] if b[1] else 0:6.3f}')
print(f"{'TOTAL':70s} {'':11s} {tg:9.2f} {'':11s} {te:9.2f} {tg / te:6.3f}")
PY
