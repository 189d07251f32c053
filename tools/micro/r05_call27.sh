#!/bin/bash
# call 27: conv_pw_h2_kernel with the 64-column tile and the implicit-GEMM loader (3x3 / strided convolutions, one or two tensors)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c27; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_winograd.py -m gpu -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; }
timeout -k 10 200 python tools/per_launch.py $O/per_launch.csv 7 > $O/per_launch.txt 2>&1; head -10 $O/per_launch.txt | tail -8
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'], [ (k['kernel'][-30:], k['ms_per_step']) for k in r['by_kernel'][:5]])" || tail -5 $O/bench.err; cp $O/last.json "$O/$(echo $1 | tr ' =' '__').json"; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=h2 ""
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_e2e.py tests/test_hip_stages.py -m gpu -x -q > $O/tests_e2e.log 2>&1; rc=$?; echo "parity / e2e / stages tests rc $rc"; tail -3 $O/tests_e2e.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_e2e.log | head -30; exit 1; }
