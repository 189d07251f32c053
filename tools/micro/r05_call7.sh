#!/bin/bash
# call 7: multi-workgroup ground-truth RLE, strided shortcuts of layer2.0 / layer3.0 inside conv3 (A/B), tests
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c7; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_conv.py tests/test_hip_mask.py tests/test_hip_parity.py tests/test_hip_e2e.py -m gpu -x -q -k "not cfg4" > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -2 $O/tests.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $O/tests.log | head -20; exit 1; }
one() { env $1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline $2 2>/dev/null | grep '^{' | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$1 $2]', round(d['value'],1), 'img/s', round(d['ms_per_step'],3), 'ms  frac', r['frac'], 'conv ms', r['all_conv_launches']['ms_per_step'], r['all_conv_launches']['launches_per_step'])"; }
for i in 1 2 3; do
  one FGN_FUSED_SHORTCUT_STRIDES=1 ""
  one FGN_FUSED_SHORTCUT_STRIDES=1,2 ""
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_stats.log 2>&1
cd $GRAFT_REPO_ROOT; cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats; grep "dense_rle\|mask_to_columns\|copyBuffer" $O/kernel_stats.csv | cut -c1-160
