#!/bin/bash
# build ablated copies of the library and time shapes A (tile4) and B (tile1) with each
cd $GRAFT_REPO_ROOT
cp fgn_amd/libfgn_hip.so /tmp/lib_orig.so
for d in 0 16 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DCONV_DBG=$d -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_dbg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fgn_amd/libfgn_hip.so /tmp/conv_dbg.o fgn_amd/csrc/abi.o fgn_amd/csrc/spatial.o fgn_amd/csrc/relation.o fgn_amd/csrc/rpn_post.o fgn_amd/csrc/det_post.o fgn_amd/csrc/mask.o
  echo "== CONV_DBG=$d"
  python tools_conv_bench.py 1,4 30 2>/dev/null | grep -E "^A|^B|^C "
done
cp /tmp/lib_orig.so fgn_amd/libfgn_hip.so
