#!/bin/bash
# PMC summary (MFMA pipe busy, wait breakdown, LDS conflicts, clock) of the dominant kernel on one GEMM shape.
# usage (GPU box): bash tools/pmc_gemm.sh <shape> <out.json>
set -euo pipefail
S=${1:-sh300}; OUTJ=${2:-gpurun_out/pmc_$S.json}
: "${GRAFT_REPO_ROOT:?run on the GPU box}"
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/pmc_gemm_$S
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES -d "$OUT/p1" --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$S" 6 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD -d "$OUT/p2" --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$S" 6 > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_CYCLES -d "$OUT/p3" --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$S" 6 > "$OUT/p3.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/p4" --output-format csv -- python3 "$ROOT/tools/gemm_one.py" "$S" 6 > "$OUT/p4.log" 2>&1 || true
python3 - "$OUT" "$ROOT/$OUTJ" "$S" <<'PY'
import csv, glob, json, sys
out, dst, shape = sys.argv[1:4]
res = {'shape': shape}
for ps in ('p1', 'p2', 'p3', 'p4'):
    for f in glob.glob(f'{out}/{ps}/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'conv_' in r['Kernel_Name']]
        last = max(int(r['Dispatch_Id']) for r in rows)
        for r in rows:
            if int(r['Dispatch_Id']) == last:
                res[r['Counter_Name']] = float(r['Counter_Value'])
                res['kernel'] = r['Kernel_Name'].split('(')[0].replace('void ', '')
                for k in ('VGPR_Count', 'Accum_VGPR_Count', 'LDS_Block_Size', 'Grid_Size', 'Workgroup_Size'):
                    if k in r: res[k] = r[k]
    for f in glob.glob(f'{out}/{ps}/**/*kernel_trace.csv', recursive=True)[:1]:
        rows = [r for r in csv.DictReader(open(f)) if 'conv_' in r['Kernel_Name']]
        if rows:
            r = rows[-1]
            res[f'duration_us_{ps}'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for l in open(f'{out}/p1.log'):
    if l.startswith('flop_per_launch'): res['flop_per_launch'] = float(l.split()[1])
d = res.get('duration_us_p1', 0) * 1e-6
if d and 'flop_per_launch' in res: res['tflops_under_pmc'] = res['flop_per_launch'] / d / 1e12
# SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs... report ratios the guide defines
if 'SQ_BUSY_CU_CYCLES' in res and res.get('SQ_VALU_MFMA_BUSY_CYCLES'):
    res['mfma_busy_over_busy_cu_cycles'] = res['SQ_VALU_MFMA_BUSY_CYCLES'] / res['SQ_BUSY_CU_CYCLES']
if res.get('SQ_INSTS_MFMA') and d:
    # 4 SIMDs x 256 CUs; 64 cycles per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32 (conv_pw_persist_kernel<true>)
    cyc = 32 if '<true>' in res.get('kernel', '') else 64
    res['mfma_cycles_per_instruction'] = cyc
    res['mfma_issue_cycles_per_simd'] = res['SQ_INSTS_MFMA'] * cyc / (256 * 4)
    if 'GRBM_GUI_ACTIVE' in res:
        clk = res['GRBM_GUI_ACTIVE'] / 8 / (res.get('duration_us_p3', 0) * 1e-6) if res.get('duration_us_p3') else 0
        res['effective_clock_ghz_from_GRBM_GUI_ACTIVE'] = clk / 1e9
        res['mfma_pipe_utilisation'] = res['mfma_issue_cycles_per_simd'] / (res['GRBM_GUI_ACTIVE'] / 8)
if res.get('SQ_WAVE_CYCLES'):
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
        if k in res: res[k + '_frac_of_wave_cycles'] = res[k] / res['SQ_WAVE_CYCLES']
if res.get('TCC_HIT_sum') is not None and res.get('TCC_MISS_sum') is not None and res['TCC_HIT_sum'] + res['TCC_MISS_sum'] > 0:
    res['l2_hit_rate'] = res['TCC_HIT_sum'] / (res['TCC_HIT_sum'] + res['TCC_MISS_sum'])
if res.get('SQ_LDS_IDX_ACTIVE'):
    res['lds_bank_conflict_frac'] = res.get('SQ_LDS_BANK_CONFLICT', 0) / res['SQ_LDS_IDX_ACTIVE']
json.dump(res, open(dst, 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
