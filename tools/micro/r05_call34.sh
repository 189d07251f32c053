#!/bin/bash
# call 34: the overflow watch as eight v_max3_f32 per sixteen elements: h2 tests, the large shapes of the probe, the step
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c34; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "h2 or two_tensor or persistent or dual" > $O/tests.log 2>&1; rc=$?; echo "conv tests rc $rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -30; exit 1; }
timeout -k 10 300 python tools/x3_probe.py --reps 10 --only "wino" > $O/probe.jsonl 2> $O/probe.err
timeout -k 10 300 python tools/x3_probe.py --reps 10 --only "relq" >> $O/probe.jsonl 2>> $O/probe.err
timeout -k 10 300 python tools/x3_probe.py --reps 10 --only "sh conv" >> $O/probe.jsonl 2>> $O/probe.err
python - <<'PY'
import json
for l in open('gpurun_out/c34/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], *[f"| {k} {d[k]['us']} ({d[k]['max_err']:.1e})" for k in ('x6_bm128','h2_bm64','h2_bm128') if k in d and 'us' in d[k]])
PY
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 > $O/last.json; python -c "import sys,json; d=json.load(open('$O/last.json')); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms', r['kernel'], 'frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'])" || tail -5 $O/bench.err; }
one FGN_GEMM_MATH=h2 ""
one FGN_GEMM_MATH=x3 ""
one FGN_GEMM_MATH=h2 ""
