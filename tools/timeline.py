"""Print the kernel timeline of one steady-state episode from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# episodes start at nchw3_to_nhwc4 on the query (grid covers 800*1333)
starts = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('nchw3_to_nhwc4')]
ep = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
ep -= ep % 2
a, b = starts[ep], starts[ep + 2]
t0 = int(rows[a]['Start_Timestamp'])
qs = sorted({r['Queue_Id'] for r in rows[a:b]})
busy_end = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    gap = s - busy_end
    busy_end = max(busy_end, e)
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
    print(f"{s / 1e3:9.1f} us +{(e - s) / 1e3:8.1f}  q{qs.index(r['Queue_Id'])}  gap{gap / 1e3:7.1f}  {name:44s} grid={r['Grid_Size_X']}")
print('episode span us', (int(rows[b]['Start_Timestamp']) - t0) / 1e3)
