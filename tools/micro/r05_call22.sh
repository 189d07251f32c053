#!/bin/bash
# call 22: 16x16x32 instances as the default, 64 / 128-row tiles chosen per launch: tests, probe, step A/B against f32
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c22; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_hip_conv.py -m gpu -x -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -2 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -20; exit 1; }
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "x3" > $O/tests_conv_exp.log 2>&1; echo "x3 tests (experiments build) rc $?"; tail -1 $O/tests_conv_exp.log
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 500 python tools/x3_probe.py --reps 10 > $O/probe.jsonl 2> $O/probe.err; echo "probe rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/c22/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], *[f"| {k} {d[k]['us']}" for k in ('f32_mfma','x6_bm64','x6_bm128','x6_bm64_mfma32','x6_bm128_mfma32') if k in d and 'us' in d[k]], 'auto', d['auto_row_tile'], d['row_tiles_equal'])
PY
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'f32eq', r.get('f32_equivalent_tflops'), 'conv ms', r['all_conv_launches']['ms_per_step'], [ (k['kernel'][-22:], k['ms_per_step']) for k in r['by_kernel'][:3]])"; }
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=f32 ""
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=f32 ""
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_e2e.py -m gpu -x -q > $O/tests_e2e.log 2>&1; rc=$?; echo "e2e tests rc $rc"; tail -2 $O/tests_e2e.log
