#!/bin/bash
# call 4: the 2 x 2 of the dominant kernel in the pipelined step - {producer-wave kernel, round-4 kernel} x {launch records
# armed, not armed} - on one box, interleaved; then the driver's command
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c4; mkdir -p $O
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  frac', r['frac'], 'window', (r.get('timed_window') or {}).get('frac'))"; }
X=FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so
for i in 1 2 3; do
  one "$X FGN_PW_WS=0" ""
  one "$X FGN_PW_WS=2" ""
  one "$X FGN_PW_WS=0" "--launch-records"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.out 2>/dev/null; grep '^{' $O/bench20.out | tail -1 > $O/bench20.json
python -c "import json; d=json.load(open('$O/bench20.json')); r=d['roofline']; print('driver-style', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'frac', r['frac'], r['isolated_steps'])"
timeout -k 10 300 python bench.py --steps 60 --warmup 5 --isolated-steps 0,15,30,45,59 --no-cpu-baseline --launch-records > $O/bench_iso.out 2>/dev/null; grep '^{' $O/bench_iso.out | tail -1 > $O/bench_iso.json
python -c "import json; d=json.load(open('$O/bench_iso.json')); r=d['roofline']; print('isolated spread', [(x['step'], x['avg_launch_us'], x['frac']) for x in r['isolated_steps']], 'window', r['timed_window'])"
