#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call: bash tools/collect_round.sh r04
set -uo pipefail
R=${1:-r05}
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/collect_$R
mkdir -p "$OUT"
cd "$ROOT"
last_json() { grep '^{' "$1" | tail -1; }
# 1. the driver's command (CPU baseline + accuracy legs)
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.out" 2> "$OUT/bench_default.err"; last_json "$OUT/bench_default.out" > "$OUT/${R}_bench_line.json"
echo "default: $(cut -c1-160 "$OUT/${R}_bench_line.json")"
# 1b. where in the timed window the isolated instrumented step sits: step 0 (the first work after the barrier's device
# synchronisation - what rounds 1-4 reported), steps in the middle and the last one (the default), one run
python bench.py --gpus 1 --steps 60 --warmup 5 --isolated-steps 0,15,30,45,59 --launch-records --no-cpu-baseline > "$OUT/b.out" 2> "$OUT/bench_iso.err"; last_json "$OUT/b.out" > "$OUT/${R}_bench_isolated_spread.json"
python - "$OUT/${R}_bench_isolated_spread.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))['roofline']
print('isolated steps:', [(x['step'], x['avg_launch_us'], x['frac']) for x in r['isolated_steps']], 'window', (r.get('timed_window') or {}).get('frac'))
PY
# 2. the other configs of BASELINE.json
for w in "cfg4 4" "cfg4 8" "cfg5 1" "cfg1 1" "cfg2 1"; do set -- $w
  python bench.py --workload $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/b.out" 2> "$OUT/bench_$1.err"; last_json "$OUT/b.out" > "$OUT/${R}_bench_$1_b$2.json"
  echo "$1 b$2: $(cut -c1-160 "$OUT/${R}_bench_$1_b$2.json")"
done
# 3. rocprofv3 kernel stats + PMC traffic of the default command, kernel stats of cfg4 with 8 episodes per step
bash tools/profile_round.sh "$R" > "$OUT/profile_round.log" 2>&1; tail -1 "$OUT/profile_round.log" | cut -c1-200
cp "$ROOT/gpurun_out/prof_$R/${R}_kernel_stats.csv" "$ROOT/gpurun_out/prof_$R/${R}_conv_traffic.json" "$OUT/" 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && rm -rf "$OUT/stats_b8" && rocprofv3 --kernel-trace --stats -d "$OUT/stats_b8" --output-format csv -- python3 "$ROOT/bench.py" --workload cfg4 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_b8_stats.log" 2>&1 )
cp "$OUT"/stats_b8/*/*kernel_stats.csv "$OUT/${R}_kernel_stats_cfg4_b8.csv" 2>/dev/null; rm -rf "$OUT/stats_b8"
# 4. multi-rank rehearsal on the one GPU (gloo, 5 ranks: the box allows 6 GPU processes)
FGN_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 5 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/r5.out" 2> "$OUT/r5.err"; last_json "$OUT/r5.out" > "$OUT/${R}_rehearsal_5rank_gloo.json"
echo "5 ranks: $(cut -c1-160 "$OUT/${R}_rehearsal_5rank_gloo.json")"
# 5. per-launch roofline table of one episode, in-kernel clock of the large GEMMs under sustained load
timeout -k 10 200 python tools/per_launch.py "$OUT/${R}_per_launch.csv" 7 > "$OUT/per_launch.txt" 2>&1; tail -14 "$OUT/per_launch.txt"
timeout -k 10 200 tools/micro/gemm_clock 2.5 0,2001 > "$OUT/${R}_gemm_clock.jsonl" 2> "$OUT/gemm_clock.err"; cut -c1-260 "$OUT/${R}_gemm_clock.jsonl"
ls "$OUT"
