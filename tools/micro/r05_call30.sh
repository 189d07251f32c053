#!/bin/bash
# call 30: the 64-episode accuracy run on the final tree
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c30; mkdir -p $O
timeout -k 10 1100 python bench.py --accuracy-episodes 64 --steps 500 --warmup 10 > $O/acc64.out 2> $O/acc64.err; echo "acc rc $?"
grep '^{' $O/acc64.out | tail -1 > $O/r05_accuracy_64.json
python -c "
import json; d=json.load(open('$O/r05_accuracy_64.json')); print(round(d['value'],1), d['gemm_math'], d['matched_pair_maxima'], d['trained_heads'].get('ap50_vs_ground_truth'), d['hip_detections_scored_against_cpu_detections'])"
