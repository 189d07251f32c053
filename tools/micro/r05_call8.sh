#!/bin/bash
# call 8: full GPU suite on the final tree (incl. the graph-key and launch-record tests), then one driver-style bench
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c8; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -30; exit 1; }
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c8/bench_default.json') if l.startswith('{')][-1])
print(round(d['value'],1), d['ms_per_step'], d['roofline']['frac'])
PY
