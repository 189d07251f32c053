#!/bin/bash
set -euo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports GRAFT_REPO_ROOT)}"
# usage: tools_pmc.sh <letter> <tile> ; writes gpurun_out/pmc_<letter>_<tile>.txt
L=$1; T=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${L}_${T}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES -d $OUT/p1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_one.py $L $T 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $OUT/p2 --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_one.py $L $T 6 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_MISC -d $OUT/p3 --output-format csv -- python $GRAFT_REPO_ROOT/tools/conv_one.py $L $T 6 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    for f in glob.glob('$OUT/%s/*/*counter_collection.csv' % p):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_igemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            print(f'{k:32s} n={len(v):3d} last={v[-1]:.4g}')
    for f in glob.glob('$OUT/%s/*/*kernel_trace.csv' % p)[:1]:
        rows=[r for r in csv.DictReader(open(f)) if 'conv_igemm' in r['Kernel_Name']]
        if rows:
            r=rows[-1]; print('  dur_us', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'vgpr', r.get('VGPR_Count'), 'accum', r.get('Accum_VGPR_Count'), 'lds', r.get('LDS_Block_Size'), 'grid', r.get('Grid_Size_X'), 'wg', r.get('Workgroup_Size_X'))
PY
