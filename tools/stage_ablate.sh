#!/bin/bash
# usage: tools/stage_ablate.sh -DCONV_DMA_STAGES=3 ... : rebuild conv_igemm.hip per flag set and run the split sweep
cd $GRAFT_REPO_ROOT
cp fgn_amd/libfgn_hip.so /tmp/lib_orig.so
OBJS=$(ls fgn_amd/csrc/*.o | grep -v conv_igemm)
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $d -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_dbg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fgn_amd/libfgn_hip.so /tmp/conv_dbg.o $OBJS
  echo "== $d"
  python tools/split_sweep.py 2>/dev/null | grep -E "columns|auto|=1 |=2 "
done
cp /tmp/lib_orig.so fgn_amd/libfgn_hip.so
