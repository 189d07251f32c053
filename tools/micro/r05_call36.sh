#!/bin/bash
# call 36: the library without its leftover environment switches: the kernels they sat in (transforms, RoIAlign, proposals, TN GEMM) and the smoke
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c36; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_winograd.py tests/test_hip_stages.py tests/test_hip_train.py -m gpu -x -q -k "not full_width and not cfg3_size and not working_detector" > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -30; exit 1; }
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
