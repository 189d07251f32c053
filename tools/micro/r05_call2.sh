#!/bin/bash
# call 2: rest of the GPU tests, producer-wave kernel on the GEMM harness, bench with the new instrumentation, A/B in the pipelined step
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c2; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_hip_e2e.py tests/test_hip_mask.py tests/test_hip_norm.py tests/test_hip_parity.py tests/test_hip_stages.py tests/test_hip_train.py tests/test_hip_winograd.py tests/test_integration_doc.py tests/test_hip_conv.py -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
timeout -k 10 120 tools/micro/gemm_clock 2.0 0,2001 > $O/gemm_clock_ws.jsonl 2> $O/gemm_clock_ws.err; echo "gemm_clock rc $?"; cut -c1-200 $O/gemm_clock_ws.jsonl
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench1.out 2> $O/bench1.err; echo "bench rc $?"; grep '^{' $O/bench1.out | tail -1 > $O/bench1.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/c2/bench1.json'))
r=d['roofline']
print('value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',r['frac'],'iso',r.get('isolated_steps'))
print('window',r.get('timed_window'))
print('classes',r.get('launch_classes'))
PY
for i in 1 2; do
  for v in 0 1; do
    FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so FGN_PW_WS=$v timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[ws=$v]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  frac', d['roofline']['frac'], d['roofline']['kernel'])"
  done
done
