#!/bin/bash
# call 28: on the final tree - the full GPU suite, the 64-episode accuracy run, the default bench line (reads the committed traffic profile)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c28; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests_all.log 2>&1; rc=$?; echo "suite rc $rc"; tail -2 $O/tests_all.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests_all.log | head -30; exit 1; }
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; grep '^{' $O/bench_default.out | tail -1 > $O/r05_bench_line.json
python -c "
import json; d=json.load(open('$O/r05_bench_line.json')); r=d['roofline']; print('default', round(d['value'],1), d['ms_per_step'], r['kernel'], r['frac'], r['f32_equivalent_tflops'], 'traffic', r['traffic'], d['matched_pair_maxima']['max_abs_dscore'])"
timeout -k 10 1000 python bench.py --accuracy-episodes 64 --steps 2000 --warmup 10 > $O/acc64.out 2> $O/acc64.err; echo "acc rc $?"
grep '^{' $O/acc64.out | tail -1 > $O/r05_accuracy_64.json
python -c "
import json; d=json.load(open('$O/r05_accuracy_64.json')); print(round(d['value'],1), d['gemm_math'], d['matched_pair_maxima'], d['trained_heads'].get('ap50_vs_ground_truth'), d['hip_detections_scored_against_cpu_detections'])"
