#!/bin/bash
# call 14: MFMA shape 16x16x32 against 32x32x16 in conv_pw_x3_kernel, variants taking turns
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c14; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "x3" > $O/tests_conv.log 2>&1; rc=$?; echo "x3 tests rc $rc"; tail -3 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -30; exit 1; }
timeout -k 10 500 python tools/x3_probe.py --reps 10 > $O/probe.jsonl 2> $O/probe.err; rc=$?; echo "probe rc $rc"; tail -3 $O/probe.err
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_x3ph.so timeout -k 10 300 python tools/x3_probe.py --reps 3 --phases --only "relq" > $O/phases.jsonl 2> $O/phases.err; echo "phases rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/c14/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], *[f"| {k} {d[k]['us']}" for k in ('f32_mfma','x6_bm64','x6_bm64_sh16','x6_bm128','x6_bm128_sh16','x9_bm64') if k in d], d['all_x6_equal'], '%.1e %.1e' % (d['x6_bm64']['max_err'], d['x6_bm64_sh16']['max_err']))
for l in open('gpurun_out/c14/phases.jsonl'):
    d=json.loads(l)
    for k in d:
        if isinstance(d[k], dict) and 'wg0_cycles_per_ktile' in d[k]: print(k, d[k]['us'], d[k]['wg0_cycles_per_ktile'])
PY
