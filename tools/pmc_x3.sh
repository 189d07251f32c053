#!/bin/bash
# PMC summary (MFMA busy, waits, LDS conflicts, instruction mix) of one conv_pw_x3_kernel / conv_pw_h2_kernel instance on one GEMM shape.
# usage (GPU box): bash tools/pmc_x3.sh <shape> <bm> <out.json> [x3 | h2]   (x3: bm 2064 / 2128 / 2129 need FGN_HIP_LIB = the experiments build)
set -euo pipefail
S=${1:-relq}; BM=${2:-64}; OUTJ=${3:-gpurun_out/pmc_x3_${S}_$BM.json}; KERN=${4:-x3}
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_x3_${S}_$BM
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES -d "$OUT/p1" --output-format csv -- python3 "$ROOT/tools/x3_one.py" "$S" "$BM" 6 "$KERN" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d "$OUT/p2" --output-format csv -- python3 "$ROOT/tools/x3_one.py" "$S" "$BM" 6 "$KERN" > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_CYCLES -d "$OUT/p3" --output-format csv -- python3 "$ROOT/tools/x3_one.py" "$S" "$BM" 6 "$KERN" > "$OUT/p3.log" 2>&1
python3 - "$OUT" "$ROOT/$OUTJ" "$S" "$BM" "$KERN" <<'PY'
import csv, glob, json, sys
out, dst, shape, bm, kern = sys.argv[1:6]
pat = 'conv_pw_' + kern
res = {'shape': shape, 'bm': int(bm)}
for ps in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{out}/{ps}/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
        if not rows: continue
        last = max(int(r['Dispatch_Id']) for r in rows)
        for r in rows:
            if int(r['Dispatch_Id']) == last:
                res[r['Counter_Name']] = float(r['Counter_Value'])
                res['kernel'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
                for k in ('VGPR_Count', 'Accum_VGPR_Count', 'LDS_Block_Size', 'Grid_Size', 'Workgroup_Size'):
                    if k in r: res[k] = r[k]
    for f in glob.glob(f'{out}/{ps}/**/*kernel_trace.csv', recursive=True)[:1]:
        rows = [r for r in csv.DictReader(open(f)) if pat in r['Kernel_Name']]
        if rows:
            r = rows[-1]
            res[f'duration_us_{ps}'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
if res.get('SQ_LDS_IDX_ACTIVE'): res['lds_bank_conflict_share'] = res.get('SQ_LDS_BANK_CONFLICT', 0.0) / res['SQ_LDS_IDX_ACTIVE']
if res.get('SQ_BUSY_CU_CYCLES') and res.get('SQ_VALU_MFMA_BUSY_CYCLES'): res['mfma_busy_over_busy_cu_cycles'] = res['SQ_VALU_MFMA_BUSY_CYCLES'] / res['SQ_BUSY_CU_CYCLES']
if res.get('SQ_WAVE_CYCLES'):
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        if k in res: res[k + '_over_wave_cycles'] = res[k] / res['SQ_WAVE_CYCLES']
json.dump(res, open(dst, 'w'), indent=1)
print(json.dumps(res))
PY
