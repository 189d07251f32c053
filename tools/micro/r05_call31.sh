#!/bin/bash
# call 31: h2 tile rule from 55 % real columns (the 76-channel RPN head leaves the f32 pipe): tests, per-launch table, the step
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c31; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_stages.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "conv / stages tests rc $rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -30; exit 1; }
timeout -k 10 200 python tools/per_launch.py $O/per_launch.csv 7 > $O/per_launch.txt 2>&1; head -12 $O/per_launch.txt | tail -10
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'])" || tail -5 $O/bench.err; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=h2 ""
timeout -k 10 600 python -m pytest tests/test_hip_e2e.py -m gpu -x -q -k "cfg3 or tolerance or cfg5" > $O/tests_e2e.log 2>&1; echo "e2e subset rc $?"; tail -2 $O/tests_e2e.log
