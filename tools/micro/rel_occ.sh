#!/bin/bash
# relation_head_kernel duration (rocprofv3 kernel trace) for the built library and an alternative build (FGN_HIP_LIB)
cd /tmp && export TMPDIR=/tmp
for lib in "" "$GRAFT_REPO_ROOT/fgn_amd/libfgn_hip_occ4.so"; do
  rm -rf /tmp/relp; FGN_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/relp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/rel_ab.py > /dev/null 2>&1
  echo "lib=[$lib]"; grep -h "relation_head\|relation_fc" /tmp/relp/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,200-
  grep -h "relation_head" /tmp/relp/*/*kernel_stats.csv | awk -F, '{print "  avg ns", $(NF-4), "min", $(NF-2)}'
done
