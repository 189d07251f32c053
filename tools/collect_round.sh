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
# 1a. the same command with the GEMM-shaped launches on conv_pw_x3_kernel (FGN_GEMM_MATH=x3) and on the f32-input MFMA kernels
# (FGN_GEMM_MATH=f32), and all three over 100 steps, arms alternating
FGN_GEMM_MATH=x3 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/b.out" 2> "$OUT/bench_x3.err"; last_json "$OUT/b.out" > "$OUT/${R}_bench_line_x3.json"
echo "x3: $(cut -c1-160 "$OUT/${R}_bench_line_x3.json")"
FGN_GEMM_MATH=f32 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/b.out" 2> "$OUT/bench_f32.err"; last_json "$OUT/b.out" > "$OUT/${R}_bench_line_f32_mfma.json"
echo "f32 MFMA: $(cut -c1-160 "$OUT/${R}_bench_line_f32_mfma.json")"
rm -f "$OUT/${R}_ab_gemm_math_100steps.jsonl"
for m in h2 x3 f32 h2 x3 f32; do
  FGN_GEMM_MATH=$m python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'gemm_math': d['gemm_math'], 'steps': d['steps'], 'img_per_s': round(d['value'],1), 'ms_per_step': round(d['ms_per_step'],3), 'kernel': d['roofline']['kernel'], 'frac': d['roofline']['frac'], 'f32_equivalent_tflops': d['roofline']['f32_equivalent_tflops']}))" | tee -a "$OUT/${R}_ab_gemm_math_100steps.jsonl"
done
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
# 5. per-launch roofline table of one episode, the three arithmetics
timeout -k 10 200 python tools/per_launch.py "$OUT/${R}_per_launch.csv" 7 > "$OUT/per_launch.txt" 2>&1; tail -14 "$OUT/per_launch.txt"
FGN_GEMM_MATH=x3 timeout -k 10 200 python tools/per_launch.py "$OUT/${R}_per_launch_x3.csv" 7 > "$OUT/per_launch_x3.txt" 2>&1; head -8 "$OUT/per_launch_x3.txt" | tail -6
FGN_GEMM_MATH=f32 timeout -k 10 200 python tools/per_launch.py "$OUT/${R}_per_launch_f32_mfma.csv" 7 > "$OUT/per_launch_f32.txt" 2>&1; head -8 "$OUT/per_launch_f32.txt" | tail -6
# 6. conv_pw_h2_kernel / conv_pw_x3_kernel beside the f32 MFMA kernels on the GEMM shapes of an episode (variants take turns)
timeout -k 10 500 python tools/x3_probe.py --reps 10 > "$OUT/${R}_h2_probe.jsonl" 2> "$OUT/h2_probe.err"; echo "h2 / x3 probe rc $?"
# 7. one training step (forward_train + backward of the heads + Adagrad + re-pack; the frozen backbone on the default arithmetic)
timeout -k 10 400 python tools/train_bench.py --steps 10 --out "$OUT/${R}_train_step.json" > "$OUT/train_bench.log" 2>&1; tail -2 "$OUT/train_bench.log" | cut -c1-300
ls "$OUT"
