#!/bin/bash
# The library and the native GEMM harness WITH the kernels that were measured and not adopted (Stream-K, the generalised
# persistent kernel, the tuning entry point fgn_conv2d_tune): tools/micro/conv_pw_experiments.inc, compiled only here.
#   tools/micro/libfgn_hip_exp.so   load with FGN_HIP_LIB=tools/micro/libfgn_hip_exp.so (same C-ABI + fgn_conv2d_tune)
#   tools/micro/gemm_clock          in-kernel clock / workgroup-span stamps of one GEMM shape per kernel variant
set -euo pipefail
cd "$(dirname "$0")/../.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DFGN_EXPERIMENTS -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-function -Wno-inline-asm"
hipcc $F -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_igemm_exp.o
objs=/tmp/conv_igemm_exp.o
for f in abi spatial norm winograd relation rpn_post det_post mask train train_bwd; do
  [ -f fgn_amd/csrc/$f.o ] || python -m fgn_amd.build
  objs="$objs fgn_amd/csrc/$f.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/libfgn_hip_exp.so $objs
hipcc $F -DCONV_CLOCK_STAMPS tools/micro/gemm_clock.hip -o tools/micro/gemm_clock
echo built tools/micro/libfgn_hip_exp.so tools/micro/gemm_clock
