#!/bin/bash
# libfgn_hip.so with the phase clocks of conv_pw_x3_kernel compiled in (-DX3_PHASES) and the kernel instances that were
# measured and not chosen (-DFGN_EXPERIMENTS): tools/micro/libfgn_hip_x3ph.so,
# loaded by tools/x3_probe.py --phases through FGN_HIP_LIB.
set -euo pipefail
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DX3_PHASES -DFGN_EXPERIMENTS -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-function -Wno-inline-asm"
hipcc $F -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_igemm_x3ph.o
objs=/tmp/conv_igemm_x3ph.o
for f in abi spatial norm winograd relation rpn_post det_post mask train train_bwd; do
  [ -f fgn_amd/csrc/$f.o ] || python -m fgn_amd.build
  objs="$objs fgn_amd/csrc/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libfgn_hip_x3ph.so $objs
echo built tools/micro/libfgn_hip_x3ph.so
