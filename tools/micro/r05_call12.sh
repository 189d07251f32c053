#!/bin/bash
# call 12: conv_pw_x3_kernel wired into the detector: conv / parity / e2e tests, then the step with x3 against f32 MFMA
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c12; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'f32eq', r.get('f32_equivalent_tflops'), 'conv ms', r['all_conv_launches']['ms_per_step'], r['all_conv_launches']['launches_per_step'])"; }
one FGN_GEMM_MATH=x3 ""; tail -3 $O/bench.err
one FGN_GEMM_MATH=f32 ""
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=f32 ""
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_e2e.py -m gpu -x -q > $O/tests_e2e.log 2>&1; rc=$?; echo "e2e tests rc $rc"; tail -3 $O/tests_e2e.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_e2e.log | head -30; exit 1; }
