#!/bin/bash
# call 32: the full GPU suite and the smoke on the final tree
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c32; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > $O/tests_all.log 2>&1; rc=$?; echo "suite rc $rc"; tail -18 $O/tests_all.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_all.log | head -30; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
