#!/bin/bash
# Produce the rocprofv3 evidence of one round: kernel stats of the bench command + PMC traffic passes.
# usage (on the GPU box): bash tools/profile_round.sh r03
# (the PMC passes serialise kernels anyway: they run the eager single-stream form of the same episodes; per-launch
# traffic of a kernel does not depend on how it was launched)
set -euo pipefail
R=${1:-r03}
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$R
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- python3 "$ROOT/bench.py" --steps 40 --warmup 3 --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" --output-format csv -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-graphs --streams 1 --inflight 1 > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" --output-format csv -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-graphs --streams 1 --inflight 1 > "$OUT/bench_write.log" 2>&1
cp "$OUT"/stats/*/*kernel_stats.csv "$OUT/${R}_kernel_stats.csv"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/fetch" "$OUT/write" "$OUT/${R}_conv_traffic.json"
tail -1 "$OUT/bench_stats.log" | cut -c1-400
