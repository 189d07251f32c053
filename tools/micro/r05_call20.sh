#!/bin/bash
# call 20: PMC counters of the x3 kernel on the relation GEMM: 32x32x16 and 16x16x32 MFMA instances, 64-row tile
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c20
timeout -k 10 300 bash tools/pmc_x3.sh relq 64 gpurun_out/c20/pmc_x3_relq_64.json > gpurun_out/c20/a.log 2>&1; echo "rc $?"; tail -1 gpurun_out/c20/a.log | cut -c1-1500
FGN_HIP_LIB=$PWD/tools/micro/libfgn_hip_exp.so timeout -k 10 300 bash tools/pmc_x3.sh relq 1064 gpurun_out/c20/pmc_x3_relq_1064.json > gpurun_out/c20/b.log 2>&1; echo "rc $?"; tail -1 gpurun_out/c20/b.log | cut -c1-1500
