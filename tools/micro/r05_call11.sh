#!/bin/bash
# call 11: phase clocks of conv_pw_x3_kernel
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c11; mkdir -p $O
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_x3ph.so timeout -k 10 300 python tools/x3_probe.py --reps 10 --phases > $O/phases.jsonl 2> $O/phases.err; rc=$?; echo "phases rc $rc"; tail -3 $O/phases.err
python - <<'PY'
import json
for l in open('gpurun_out/c11/phases.jsonl'):
    d=json.loads(l)
    print(d['shape'], 'f32', d['f32_mfma']['us'])
    for k in ('x6_bm128','x6_bm64','x9_bm64'):
        if k in d:
            print('   ', k, d[k]['us'], '%.1e' % d[k]['max_err'], d[k].get('wg0_cycles_per_ktile'), d[k].get('wg1_cycles_per_ktile',{}).get('wait_barrier'))
PY
