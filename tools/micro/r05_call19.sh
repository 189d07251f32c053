#!/bin/bash
# call 19: the bench lines that read the committed traffic profile / the launch records, on the final tree
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c19; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; grep '^{' $O/bench_default.out | tail -1 > $O/r05_bench_line.json
python bench.py --gpus 1 --steps 60 --warmup 5 --isolated-steps 0,15,30,45,59 --launch-records --no-cpu-baseline > $O/b.out 2> $O/bench_iso.err; grep '^{' $O/b.out | tail -1 > $O/r05_bench_isolated_spread.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/c19/r05_bench_line.json')); r=d['roofline']
print('default', round(d['value'],1), d['ms_per_step'], r['kernel'], r['frac'], r['f32_equivalent_tflops'], 'traffic', r['traffic'])
x=json.load(open('gpurun_out/c19/r05_bench_isolated_spread.json')); r=x['roofline']
print('iso', round(x['value'],1), [(i['step'],i['avg_launch_us'],i['frac']) for i in r['isolated_steps']])
tw=r['timed_window']; print({k:tw[k] for k in tw if k!='what'} if tw else None)
PY
