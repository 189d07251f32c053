#!/bin/bash
# call 35: final tree (overflow watch on v_max3_f32): parity / stages / end-to-end subset, the default bench line, the per-launch table
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c35; mkdir -p $O
timeout -k 10 330 python -m pytest tests/test_hip_parity.py tests/test_hip_stages.py tests/test_hip_e2e.py -m gpu -x -q -k "not cfg4_batched and not full_width and not half_width" > $O/tests.log 2>&1; rc=$?; echo "parity / stages / e2e subset rc $rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -30; exit 1; }
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.out 2> $O/bench_default.err; grep '^{' $O/bench_default.out | tail -1 > $O/r05_bench_line.json
python -c "
import json; d=json.load(open('$O/r05_bench_line.json')); r=d['roofline']; print('default', round(d['value'],1), d['ms_per_step'], r['kernel'], r['frac'], r['f32_equivalent_tflops'], 'traffic', r['traffic'], r['avg_launch_us'], d['matched_pair_maxima'])"
timeout -k 10 100 python tools/per_launch.py $O/r05_per_launch.csv 7 > $O/per_launch.txt 2>&1; head -5 $O/per_launch.txt | tail -3
