#!/bin/bash
# Does a rank that runs a step late stall the others' COMPUTE?  Five ranks on the one GPU (gloo rehearsal), every rank is
# late by J ms once every `world` steps (FGN_BENCH_JITTER_MS), the per-step collective either on the communication
# stream (default) or on the caller stream (FGN_BENCH_GATHER_ON_COMPUTE=1).  With the collective in the compute
# stream every rank's delay is paid by everybody (K/world * world * J extra per rank); off it, delays are absorbed by the
# queue depth and a rank pays only its own.  usage: bash tools/rehearse_jitter.sh r04 [J_ms]
set -uo pipefail
R=${1:-r04}; J=${2:-60}
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/jitter_$R; mkdir -p "$OUT"; cd "$GRAFT_REPO_ROOT"
last_json() { grep '^{' "$1" | tail -1; }
run() {  # name, env...
  name=$1; shift
  env FGN_BENCH_BACKEND=gloo "$@" timeout -k 10 400 python bench.py --gpus 5 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/$name.out" 2> "$OUT/$name.err" || { echo "$name failed"; tail -5 "$OUT/$name.err"; return 1; }
  last_json "$OUT/$name.out" > "$OUT/$name.json"
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d['config']
print(f"{sys.argv[2]:28s} value {d['value']:7.1f} img/s  ms/step {d['ms_per_step']:7.2f}  per-rank {c['per_rank_ms_per_step']['all']}  gather on {c['gather_stream']}  identical {d.get('gather', {}).get('rank0_episode_identical_to_local_result')}")
PY
}
run comm_nojitter && run compute_nojitter FGN_BENCH_GATHER_ON_COMPUTE=1 && \
run comm_jitter FGN_BENCH_JITTER_MS=$J && run compute_jitter FGN_BENCH_JITTER_MS=$J FGN_BENCH_GATHER_ON_COMPUTE=1
python - "$OUT" "$R" "$J" <<'PY'
import json, sys, os
out, r, j = sys.argv[1:4]
res = {}
for n in ('comm_nojitter', 'compute_nojitter', 'comm_jitter', 'compute_jitter'):
    p = os.path.join(out, n + '.json')
    if os.path.exists(p):
        d = json.load(open(p)); c = d['config']
        res[n] = dict(ms_per_step=d['ms_per_step'], per_rank_ms=c['per_rank_ms_per_step']['all'], gather_stream=c['gather_stream'],
                      jitter_ms=c.get('jitter_ms_rehearsal'), identical=d.get('gather', {}).get('rank0_episode_identical_to_local_result'))
json.dump(dict(what='5 ranks share one MI355X (gloo rehearsal, 20 steps): each rank is late by jitter_ms once every 5 steps; '
               'collective on the communication stream vs on the caller stream', jitter_ms=float(j), runs=res),
          open(os.path.join(out, f'{r}_rehearsal_jitter_5rank.json'), 'w'), indent=1)
PY
