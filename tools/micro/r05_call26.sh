#!/bin/bash
# call 26: conv_pw_h2_kernel as the default (own scale everywhere, no record plumbing): conv / winograd tests, per-launch table, step A/B
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c26; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_winograd.py -m gpu -x -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
timeout -k 10 200 python tools/per_launch.py $O/per_launch.csv 7 > $O/per_launch.txt 2>&1; head -8 $O/per_launch.txt | tail -6
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'], [ (k['kernel'][-26:], k['ms_per_step']) for k in r['by_kernel'][:4]])" || tail -5 $O/bench.err; cp $O/last.json "$O/$(echo $1 | tr ' =' '__').json"; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; grep '^{' $O/bench_default.out | tail -1 > $O/bench_line.json
python -c "
import json; d=json.load(open('$O/bench_line.json')); r=d['roofline']; print('default line', round(d['value'],1), d['ms_per_step'], r['kernel'], r['frac'], r['f32_equivalent_tflops'], d['matched_pair_maxima'], d['cpu_baseline']['value'])"
