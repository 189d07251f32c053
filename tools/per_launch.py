"""Per-launch roofline table of one cfg3 episode (profiles/rNN_per_launch.csv): every convolution-family launch in
launch order - kind, device kernel, GEMM shape (groups, rows, N, K), 64x64-tile equivalents, duration, issued GFLOP,
TF/s (f32 products), fraction of the 157.3 TF/s f32 MFMA peak and of the pipe the kernel runs on.  The episode runs eagerly on ONE stream with the support branch on the
caller's stream too, so every launch has the chip to itself (the durations a rocprofv3 kernel trace shows); each launch is
stamped by its own start / stop events (fgn_profile_next_launch); medians over `reps` episodes.
usage: per_launch.py out.csv [reps]"""
import csv, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
out_csv = sys.argv[1] if len(sys.argv) > 1 else 'per_launch.csv'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
model.use_side_stream = False
b = make_batch(0, 1, **CONFIGS['cfg3'])
e = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
e['img_shape'] = e['img_shape'].cpu()
runs = []
for it in range(reps + 2):
    prof = ops.ConvProfile()
    ops.PROFILE = prof if it >= 2 else None
    model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'])
    ops.PROFILE = None
    torch.cuda.synchronize()
    if it >= 2:
        runs.append([(r, r['e0'].elapsed_time(r['e1']) * 1e3) for r in prof])
n = len(runs[0])
assert all(len(r) == n for r in runs)
rows, tot_us, tot_fl = [], 0.0, 0.0
for i in range(n):
    rec = runs[0][i][0]
    us = statistics.median(r[i][1] for r in runs)
    nd, n_img = rec['n_img_dev'], rec['n_img']
    cnt = n_img if nd is None else min(n_img, int(nd.item()))
    fl = rec['flop_issued'] * cnt
    g, rows_per, N, K = rec.get('gemm', (0, 0, 0, 0))
    M = rows_per * (cnt if rec['kind'] in ('conv', 'wg_gemm') else 1)
    tiles = g * ((M + 63) // 64) * ((N + 63) // 64) if g else 0
    # frac: f32 products per second against the f32-input MFMA peak (conv_pw_x3_kernel can exceed 1: it does not run on
    # that pipe); frac_of_pipe: against the pipe the kernel runs on (x3: six bf16 MFMA products per f32 product / 2500)
    x3 = rec.get('math') == 'x3'
    terms = {'x3': 6, 'h2': 3}.get(rec.get('math'))
    rows.append(dict(i=i, kind=rec['kind'], kernel=rec['kernel'], groups=g, M=M, N=N, K=K, tiles64=tiles, us=round(us, 1),
                     gflop=round(fl / 1e9, 2), tflops=round(fl / us / 1e6, 1) if fl else '',
                     frac=round(fl / us / 1e6 / 157.3, 3) if fl else '',
                     frac_of_pipe=(round(fl / us / 1e6 * (terms / 2500.0 if terms else 1 / 157.3), 3) if fl else ''),
                     layer_shape='x'.join(str(v) for v in rec['shape'])))
    tot_us += us
    tot_fl += fl
with open(out_csv, 'w', newline='') as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0]))
    w.writeheader()
    w.writerows(rows)
print(f'{n} launches, {tot_us / 1e3:.3f} ms of kernels, {tot_fl / 1e9:.1f} GFLOP issued, {tot_fl / tot_us / 1e6:.1f} TF/s over the conv family')
gem = [r for r in rows if r['gflop']]
by = {}
for r in gem:
    k = by.setdefault(r['kernel'], [0, 0.0, 0.0])
    k[0] += 1; k[1] += r['us']; k[2] += r['gflop']
for k, (c, us, gf) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f'{k:58s} {c:3d} launches {us:8.1f} us {gf:7.1f} GFLOP {gf / us * 1e3:6.1f} TF/s  {gf / us * 1e3 / 157.3:.3f}')
print('worst by time x (1 - frac):')
for r in sorted(gem, key=lambda r: -r['us'] * (1 - r['frac']))[:12]:
    print(f"  #{r['i']:3d} {r['kind']:8s} g{r['groups']:2d} M{r['M']:6d} N{r['N']:5d} K{r['K']:5d} {r['us']:7.1f} us {r['tflops']:6.1f} TF/s {r['frac']:.3f}  {r['kernel']}")
