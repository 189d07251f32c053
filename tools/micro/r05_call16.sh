#!/bin/bash
# call 16: per-launch routing rule (x3 only where it wins): conv tests, per-launch table, step A/B, full GPU suite
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c16; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -2 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
timeout -k 10 200 python tools/per_launch.py $O/per_launch_x3.csv 7 > $O/per_launch_x3.txt 2>&1; echo "per_launch rc $?"; head -9 $O/per_launch_x3.txt | cut -c1-150
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'f32eq', r.get('f32_equivalent_tflops'), 'conv ms', r['all_conv_launches']['ms_per_step'], r['all_conv_launches']['launches_per_step'])"; }
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=f32 ""
one FGN_GEMM_MATH=x3 ""
cp $O/last.json $O/bench_x3_100.json
timeout -k 10 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default bench rc $?"; python -c "
import json; d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); r=d['roofline']; print('default', round(d['value'],1), d['ms_per_step'], r['kernel'], r['frac'], r['f32_equivalent_tflops'], d['matched_pair_maxima'])"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_hip_conv.py > $O/tests_all.log 2>&1; rc=$?; echo "suite rc $rc"; tail -3 $O/tests_all.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_all.log | head -30; exit 1; }
