#!/bin/bash
# call 18: the pipeline's knobs again now that the GEMMs are shorter: phase mark, episodes in flight (100-step windows)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c18; mkdir -p $O
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>$O/bench.err | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  lat p50', d['config']['episode_latency_ms']['p50'])" | tee -a $O/sweep.txt; }
one FGN_BENCH_PHASE=rpn ""
one FGN_BENCH_PHASE=layer3 ""
one FGN_BENCH_PHASE=rpn_conv ""
one FGN_BENCH_PHASE=proposals ""
one FGN_BENCH_PHASE=off ""
one FGN_BENCH_PHASE=rpn "--inflight 4"
one FGN_BENCH_PHASE=rpn "--inflight 2"
one FGN_BENCH_PHASE=rpn ""
