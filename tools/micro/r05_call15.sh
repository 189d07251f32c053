#!/bin/bash
# call 15: per-launch tables of one episode with the GEMMs on conv_pw_x3_kernel and on the f32 MFMA kernels (one box)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/c15; mkdir -p $O
FGN_GEMM_MATH=x3 timeout -k 10 200 python tools/per_launch.py $O/per_launch_x3.csv 7 > $O/per_launch_x3.txt 2>&1; echo "x3 rc $?"; head -12 $O/per_launch_x3.txt | cut -c1-160
FGN_GEMM_MATH=f32 timeout -k 10 200 python tools/per_launch.py $O/per_launch_f32.csv 7 > $O/per_launch_f32.txt 2>&1; echo "f32 rc $?"; head -10 $O/per_launch_f32.txt | cut -c1-160
python - <<'PY'
import csv
a=list(csv.DictReader(open('gpurun_out/c15/per_launch_x3.csv'))); b=list(csv.DictReader(open('gpurun_out/c15/per_launch_f32.csv')))
assert len(a)==len(b)
tot=[0,0]
for x,y in zip(a,b):
    if x['kernel']!=y['kernel']:
        print(f"#{x['i']:>3s} {x['kind']:8s} g{x['groups']:>2s} M{x['M']:>6s} N{x['N']:>5s} K{x['K']:>5s}  f32 {float(y['us']):7.1f}  x3 {float(x['us']):7.1f}  {float(y['us'])/float(x['us']):.2f}x   {y['kernel'][:44]}")
    tot[0]+=float(x['us']); tot[1]+=float(y['us'])
print('total us x3', round(tot[0]), 'f32', round(tot[1]))
PY
