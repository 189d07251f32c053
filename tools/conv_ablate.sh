#!/bin/bash
# usage: tools/conv_ablate.sh "<grep pattern of conv_bench rows>" "<tiles>" -DCONV_DBG=1 -DCONV_DBG=8 ...
cd $GRAFT_REPO_ROOT
PAT="$1"; TILES="$2"; shift 2
cp fgn_amd/libfgn_hip.so /tmp/lib_orig.so
OBJS=$(ls fgn_amd/csrc/*.o | grep -v conv_igemm)
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $d -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_dbg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fgn_amd/libfgn_hip.so /tmp/conv_dbg.o $OBJS
  echo "== $d"
  python tools/conv_bench.py $TILES 30 2>/dev/null | grep -E "$PAT"
done
cp /tmp/lib_orig.so fgn_amd/libfgn_hip.so
