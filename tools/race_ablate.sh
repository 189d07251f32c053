#!/bin/bash
# usage: tools/race_ablate.sh N -DCONV_DBG=64 ...   : rebuild conv_igemm.hip with each flag set and run the race probe
cd $GRAFT_REPO_ROOT
N=$1; shift
cp fgn_amd/libfgn_hip.so /tmp/lib_orig.so
OBJS=$(ls fgn_amd/csrc/*.o | grep -v conv_igemm)
for d in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $d -c fgn_amd/csrc/conv_igemm.hip -o /tmp/conv_dbg.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o fgn_amd/libfgn_hip.so /tmp/conv_dbg.o $OBJS
  echo "== $d"
  python tools/conv_race_probe.py $N 2>/dev/null | grep -E "sh300|run"
done
cp /tmp/lib_orig.so fgn_amd/libfgn_hip.so
