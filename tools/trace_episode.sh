#!/bin/bash
# Kernel timeline of one steady-state episode of bench.py: bash tools/trace_episode.sh <tag> [bench args]
set -euo pipefail
prev=""
TAG=${1:-tl}; shift || true
# single rank only: with --gpus N > 1 bench.py would start torch.distributed.run as a child of the PROFILED process -
# the launcher hop under the profiler's preloaded library that this pool forbids
for a in "$@"; do
    if [[ "$prev" == "--gpus" && "$a" != "1" ]] || [[ "$a" == --gpus=* && "$a" != "--gpus=1" ]]; then
        echo "trace_episode.sh: profiling is single-rank (python3 bench.py directly after --); drop --gpus" >&2; exit 2
    fi
    prev=$a
done
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/kt" --output-format csv -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$OUT/bench.log" 2>&1
python3 "$ROOT/tools/timeline.py" "$OUT/kt" > "$ROOT/gpurun_out/timeline_$TAG.txt"
rm -rf "$OUT/kt"
tail -1 "$OUT/bench.log" | cut -c1-200
