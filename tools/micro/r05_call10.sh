#!/bin/bash
# call 10: conv_pw_x3_kernel with the 3-stage ring (BM 128) / 2-stage (BM 64): error and time beside the f32 MFMA kernels
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c10; mkdir -p $O
timeout -k 10 120 python tools/x3_probe.py --reps 5 --only "layer3 conv1" > $O/probe_first.jsonl 2> $O/probe_first.err; rc=$?; echo "first rc $rc"; cut -c1-600 $O/probe_first.jsonl; tail -5 $O/probe_first.err
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/x3_probe.py --reps 30 > $O/probe.jsonl 2> $O/probe.err; rc=$?; echo "probe rc $rc"; tail -3 $O/probe.err
python - <<'PY'
import json
for l in open('gpurun_out/c10/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], 'f32', d['f32_mfma']['us'], '%.1e/%.1e' % (d['f32_mfma']['max_err'], d['f32_mfma']['mean_err']),
          *[f"| {k} {d[k]['us']} {d[k]['max_err']:.1e}/{d[k]['mean_err']:.1e}" for k in ('x6_bm128','x6_bm64','x9_bm128','x9_bm64') if k in d], d.get('bm128_equals_bm64'))
PY
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests/test_hip_train.py -m gpu -x -q -k "repack" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
