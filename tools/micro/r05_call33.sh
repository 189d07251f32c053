#!/bin/bash
# call 33: SQ / LDS counters of conv_pw_h2_kernel (64-row instance on the relation GEMM, 128-row instance on the AG-RPN Winograd GEMM)
set -uo pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/c33
timeout -k 10 300 bash tools/pmc_x3.sh relq 64 gpurun_out/c33/r05_pmc_h2_relq.json h2 > gpurun_out/c33/a.log 2>&1; echo "rc $?"; tail -1 gpurun_out/c33/a.log | cut -c1-1500
timeout -k 10 300 bash tools/pmc_x3.sh agrpn 128 gpurun_out/c33/r05_pmc_h2_agrpn.json h2 > gpurun_out/c33/b.log 2>&1; echo "rc $?"; tail -1 gpurun_out/c33/b.log | cut -c1-1500
