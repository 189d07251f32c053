"""Print the kernel timeline of one steady-state episode from a rocprofv3 kernel trace CSV, per HIP queue."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for mf in glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(mf)):
        rows.append({'Kernel_Name': f"MEMCPY {r.get('Direction', '')} {int(r.get('Bytes', r.get('Size', 0)) or 0) / 1e6:.2f}MB",
                     'Start_Timestamp': r['Start_Timestamp'], 'End_Timestamp': r['End_Timestamp'],
                     'Queue_Id': 'copy', 'Grid_Size_X': '0', 'Workgroup_Size_X': '0'})
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# an episode ends with its mask_rle_kernel (one per episode, last kernel on the caller's stream)
ends = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('mask_rle_kernel')]
ep = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) // 2
a, b = ends[ep] + 1, ends[ep + 1] + 1
t0 = int(rows[ends[ep]]['End_Timestamp'])
main_q = rows[ends[ep]]['Queue_Id']
qs = sorted({r['Queue_Id'] for r in rows[a:b]}, key=lambda q: q != main_q)
busy_end = collections.defaultdict(int)
tot = collections.defaultdict(float)
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    q = r['Queue_Id']
    gap = s - busy_end[q] if busy_end[q] else 0
    busy_end[q] = max(busy_end[q], e)
    name = r['Kernel_Name'].split('(')[0].replace('void ', '')[:52]
    tot[(qs.index(q), name)] += (e - s) / 1e3
    print(f"{s / 1e3:9.1f} us +{(e - s) / 1e3:8.1f}  q{qs.index(q)}  gap{gap / 1e3:7.1f}  {name:52s} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']}")
print('episode span us', (int(rows[b - 1]['End_Timestamp']) - t0) / 1e3)
print('--- per queue / kernel totals (us)')
for (q, n), v in sorted(tot.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f'q{q} {n:52s} {v:9.1f}')
for qi in range(len(qs)):
    print(f'q{qi} total busy us', round(sum(v for (q, n), v in tot.items() if q == qi), 1))
