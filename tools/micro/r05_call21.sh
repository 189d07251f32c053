#!/bin/bash
# call 21: the 16x16x32 instances with the conflict-free k order: tests (experiments build), PMC, probe
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c21; mkdir -p $O
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 300 python -m pytest tests/test_hip_conv.py -m gpu -x -q -k "x3" > $O/tests_conv.log 2>&1; rc=$?; echo "x3 tests (experiments build) rc $rc"; tail -2 $O/tests_conv.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_conv.log | head -20; exit 1; }
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 300 bash tools/pmc_x3.sh relq 1064 $O/pmc_x3_relq_1064.json > $O/b.log 2>&1; python -c "
import json; d=json.load(open('$O/pmc_x3_relq_1064.json')); print({k:d[k] for k in ('lds_bank_conflict_share','SQ_LDS_IDX_ACTIVE','SQ_WAVE_CYCLES','duration_us_p1') if k in d})"
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 500 python tools/x3_probe.py --reps 10 > $O/probe.jsonl 2> $O/probe.err; echo "probe rc $?"
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_x3ph.so timeout -k 10 300 python tools/x3_probe.py --reps 3 --phases --only relq > $O/phases.jsonl 2> $O/phases.err
python - <<'PY'
import json
for l in open('gpurun_out/c21/probe.jsonl'):
    d=json.loads(l)
    print(d['shape'], *[f"| {k} {d[k]['us']}" for k in ('f32_mfma','x6_bm64','x6_bm64_sh16','x6_bm128','x6_bm128_sh16') if k in d and 'us' in d[k]], '%.1e %.1e' % (d['x6_bm64']['max_err'], d.get('x6_bm64_sh16',{}).get('max_err',0)))
for l in open('gpurun_out/c21/phases.jsonl'):
    d=json.loads(l)
    for k in d:
        if isinstance(d[k], dict) and 'wg0_cycles_per_ktile' in d[k]: print(k, d[k]['us'], d[k]['wg0_cycles_per_ktile'])
PY
