#!/bin/bash
# call 24: conv_pw_h2_kernel (three f16 products): its tests, the probe against x3 / f32, the step A/B h2 / x3
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c24; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "h2" -s > $O/tests_h2.log 2>&1; rc=$?; echo "h2 tests rc $rc"; tail -3 $O/tests_h2.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_h2.log | head -30; }
grep "of the range" $O/tests_h2.log
if [ -z "${SKIP_PROBE:-}" ]; then timeout -k 10 500 python tools/x3_probe.py --reps 10 > $O/probe.jsonl 2> $O/probe.err; echo "probe rc $?"; tail -3 $O/probe.err; fi
python - <<'PY'
import json
for l in open('gpurun_out/c24/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], *[f"| {k} {d[k]['us']} ({d[k]['max_err']:.1e})" for k in ('x6_bm64','x6_bm128','h2_bm64','h2_bm128','h2o_bm64','h2o_bm128','h2o_bm64_st3') if k in d and 'us' in d[k]])
PY
[ $rc -eq 0 ] || exit 1
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'], [ (k['kernel'][-26:], k['ms_per_step']) for k in r['by_kernel'][:4]], d['matched_pair_maxima'])" || tail -5 $O/bench.err; }
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_winograd.py -m gpu -x -q > $O/tests_conv.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one "FGN_GEMM_MATH=h2 FGN_H2_WG_RECORD=0" ""
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one "FGN_GEMM_MATH=h2 FGN_H2_WG_RECORD=0" ""
