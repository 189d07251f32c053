#!/bin/bash
# call 25: the step on conv_pw_h2_kernel (record / own scale for the Winograd GEMMs) against conv_pw_x3_kernel, arms alternating
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c25; mkdir -p $O
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'], [ (k['kernel'][-26:], k['ms_per_step']) for k in r['by_kernel'][:4]])" || tail -5 $O/bench.err; cp $O/last.json "$O/$(echo $1 | tr ' =' '__').json"; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one "FGN_GEMM_MATH=h2 FGN_H2_WG_RECORD=0" ""
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one "FGN_GEMM_MATH=h2 FGN_H2_WG_RECORD=0" ""
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_e2e.py tests/test_hip_stages.py -m gpu -x -q > $O/tests_e2e.log 2>&1; rc=$?; echo "parity / e2e / stages tests rc $rc"; tail -3 $O/tests_e2e.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_e2e.log | head -30; exit 1; }
