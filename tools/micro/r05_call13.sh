#!/bin/bash
# call 13: conv_pw_x3_kernel with 64 x 64 wave tiles (128-row workgroup tile, 4 waves, 2 per CU): phases + times, x3 tests
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c13; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "x3" > $O/tests_conv.log 2>&1; rc=$?; echo "x3 tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_x3ph.so timeout -k 10 300 python tools/x3_probe.py --reps 10 --phases > $O/phases.jsonl 2> $O/phases.err; rc=$?; echo "phases rc $rc"; tail -3 $O/phases.err
timeout -k 10 300 python tools/x3_probe.py --reps 30 > $O/probe.jsonl 2> $O/probe.err; rc=$?; echo "probe rc $rc"
python - <<'PY'
import json
ph={json.loads(l)['shape']:json.loads(l) for l in open('gpurun_out/c13/phases.jsonl')}
for l in open('gpurun_out/c13/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], 'f32', d['f32_mfma']['us'], *[f"| {k} {d[k]['us']}" for k in ('x6_bm64','x6_bm128','x6_bm129','x9_bm64') if k in d])
    p=ph.get(d['shape'],{})
    for k in ('x6_bm64','x6_bm128'):
        if k in p and 'wg0_cycles_per_ktile' in p[k]: print('     ', k, p[k]['wg0_cycles_per_ktile'])
PY
