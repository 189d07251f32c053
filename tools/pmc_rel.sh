#!/bin/bash
# SQ counters of relation_head_kernel at the cfg3 shape (tools/rel_ab.py): where its cycles go.  GPU box only.
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_rel
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/rel_ab.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS -d $OUT/p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/rel_ab.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAIT_INST_LDS SQ_INSTS_LDS -d $OUT/p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/rel_ab.py > /dev/null 2>&1 || true
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv' % p):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'relation_head' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            print(f'{k:32s} n={len(v):3d} median={sorted(v)[len(v)//2]:.5g}')
    for f in glob.glob('$OUT/%s/*/*kernel_trace.csv' % p)[:1]:
        rows=[r for r in csv.DictReader(open(f)) if 'relation_head' in r['Kernel_Name']]
        d=sorted((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows)
        if rows:
            r=rows[-1]; print('  dur_us median', d[len(d)//2], 'min', d[0], 'vgpr', r.get('VGPR_Count'), 'lds', r.get('LDS_Block_Size'), 'grid', r.get('Grid_Size_X'), 'wg', r.get('Workgroup_Size_X'))
PY
