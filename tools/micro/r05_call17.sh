#!/bin/bash
# call 17: 64 cfg3 episodes HIP (GEMMs on conv_pw_x3_kernel) against the oracle, off the timed path, + 2000 timed steps
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c17; mkdir -p $O
timeout -k 10 1100 python bench.py --accuracy-episodes 64 --steps 2000 --warmup 10 > $O/acc64.out 2> $O/acc64.err; echo "rc $?"
grep '^{' $O/acc64.out | tail -1 > $O/r05_accuracy_64.json
python -c "
import json; d=json.load(open('$O/r05_accuracy_64.json')); print(round(d['value'],1), d['gemm_math'], d['matched_pair_maxima'], d['trained_heads'].get('ap50_vs_ground_truth'), d['hip_detections_scored_against_cpu_detections'])"
