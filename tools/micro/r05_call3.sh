#!/bin/bash
# call 3: producer-wave kernel as the product kernel (tests, harness vs the round-4 form), packed transfers A/B,
# every unsplit point-wise launch on the persistent kernel (experiment), per-launch table
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_winograd.py tests/test_hip_parity.py tests/test_hip_mask.py tests/test_integration_doc.py tests/test_hip_e2e.py -m gpu -x -q -k "not cfg4" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 10 120 tools/micro/gemm_clock 2.0 0,2002 > $O/gemm_clock_ws.jsonl 2> $O/gemm_clock_ws.err; echo "gemm_clock rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/c3/gemm_clock_ws.jsonl'):
    d=json.loads(l); print(d['shape'][:34], 'v', d['variant'], d['us_per_launch_back_to_back'], 'us', d['frac_of_157.3'], 'clk', d['clock_ghz_median'])
PY
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$1]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  frac', r['frac'], 'window', (r.get('timed_window') or {}).get('frac'), 'conv ms', r['all_conv_launches']['ms_per_step'])"; }
for i in 1 2; do
  one FGN_PACKED_TRANSFERS=0
  one FGN_PACKED_TRANSFERS=1
done
X=FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so
for i in 1 2; do
  one "$X FGN_PW_WS=0"
  one "$X FGN_PW_WS=3"
  one "$X FGN_PW_WS=2"
done
timeout -k 10 200 python tools/per_launch.py $O/r05_per_launch.csv 7 > $O/per_launch.txt 2>&1; tail -14 $O/per_launch.txt
