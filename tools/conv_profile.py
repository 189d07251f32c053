"""Per-launch conv timing of one cfg3 episode (diagnostic)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fgn_amd import ops
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
shape = CONFIGS['cfg3']
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
b = make_batch(0, 1, **shape)
e = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
e['img_shape'] = e['img_shape'].cpu()
for it in range(3):
    prof = ops.ConvProfile()
    ops.PROFILE = prof if it == 2 else None
    model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'])
    ops.PROFILE = None
    torch.cuda.synchronize()
tot = 0
agg = {}
for rec in prof:
    ms = rec['e0'].elapsed_time(rec['e1'])
    nd, n_img = rec['n_img_dev'], rec['n_img']
    n = n_img if nd is None else min(n_img, int(nd.item()))
    fl = rec['flop_issued'] * n
    key = (rec['kind'],) + rec['shape']
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += ms; a[2] += fl
    tot += ms
print(f'total conv ms {tot:.3f}')
print(f'{"n_img,H,W,Cin,Cout,k,s":40s} {"calls":>5s} {"ms":>8s} {"GFLOP":>8s} {"TF/s":>7s} {"M":>7s}')
for key, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, n_img, H, W, cin, cout, k, s = key
    M = n_img * ((H + 2 * (k // 2) - k) // s + 1) * ((W + 2 * (k // 2) - k) // s + 1)
    print(f'{str(key):40s} {c:5d} {ms:8.3f} {fl / 1e9:8.1f} {fl / ms / 1e9:7.1f} {M:7d}')
