#!/bin/bash
cd $GRAFT_REPO_ROOT
cp fgn_amd/libfgn_hip.so /tmp/lib_orig.so
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $d -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_dbg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fgn_amd/libfgn_hip.so /tmp/conv_dbg.o fgn_amd/csrc/abi.o fgn_amd/csrc/spatial.o fgn_amd/csrc/relation.o fgn_amd/csrc/rpn_post.o fgn_amd/csrc/det_post.o fgn_amd/csrc/mask.o
  echo "== $d"
  python tools/conv_bench.py 4 30 2>/dev/null | grep -E "^D|^J|^G|^B"
done
cp /tmp/lib_orig.so fgn_amd/libfgn_hip.so
