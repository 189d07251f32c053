"""GPU occupancy of the pipelined bench from a rocprofv3 kernel trace: over the steady-state window (between the 20 %
and 90 % quantile of the kernels' start times) - fraction of the wall with at least one kernel running, mean number of
kernels running, the idle gaps, and the same split by "a large GEMM is running" (conv_pw_persist*) or not.
usage: busy.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
n = len(rows)
t0, t1 = rows[int(n * 0.2)][0], rows[int(n * 0.9)][0]
ev = []
for s, e, name in rows:
    if e <= t0 or s >= t1:
        continue
    s, e = max(s, t0), min(e, t1)
    big = name.startswith('void conv_pw_persist')
    ev.append((s, 1, big)); ev.append((e, -1, big))
ev.sort()
wall = t1 - t0
cur = curbig = 0
last = t0
busy = sumdur = bigbusy = idle_with_nothing = 0
gaps = []
for t, d, big in ev:
    dt = t - last
    if cur > 0:
        busy += dt
        sumdur += dt * cur
        if curbig > 0:
            bigbusy += dt
    elif dt > 0:
        gaps.append(dt)
    cur += d
    if big:
        curbig += d
    last = t
gaps.sort()
print(f'window {wall / 1e6:.2f} ms, {len(ev) // 2} kernels')
print(f'at least one kernel running: {busy / wall:.4f} of the wall; mean kernels running {sumdur / wall:.2f}; a persistent GEMM running {bigbusy / wall:.4f}')
print(f'idle: {sum(gaps) / wall:.4f} of the wall in {len(gaps)} gaps; median {gaps[len(gaps) // 2] / 1e3 if gaps else 0:.1f} us, p90 {gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0:.1f} us, max {gaps[-1] / 1e3 if gaps else 0:.1f} us; gaps > 20 us: {sum(g for g in gaps if g > 20000) / wall:.4f} of the wall')
