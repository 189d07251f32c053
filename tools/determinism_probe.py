"""Run the cfg3 episode several times with the debug trace on and report the first traced tensor that
is not bit-identical run to run (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
shape = CONFIGS['cfg3']
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
batch = make_batch(21, 1, **shape)
runs = []
for i in range(4):
    model.debug_trace = {}
    model.simple_test(**batch, rescale=True)
    torch.cuda.synchronize()
    tr = model.debug_trace
    flat = {k: v.clone() for k, v in tr.items() if isinstance(v, torch.Tensor)}
    for k, v in tr['per_image'][0].items():
        if isinstance(v, torch.Tensor):
            flat['img0.' + k] = v.clone()
    runs.append(flat)
n_props = int(runs[0]['n_props'][0]); n_det = int(runs[0]['img0.n_det'][0])
print('n_props', n_props, 'n_det', n_det)
for k in runs[0]:
    bad = []
    for i in range(1, 4):
        a, b = runs[0][k], runs[i][k]
        if k in ('img0.roi_feats', 'img0.Q', 'img0.cls_raw', 'img0.reg_raw', 'img0.rois'):
            n = n_props * (a.shape[0] // 300)
            a, b = a[:n], b[:n]
        if k in ('img0.mask_feats', 'img0.mask_logits', 'img0.mask_prob', 'img0.masks', 'img0.det', 'img0.lab'):
            a, b = a[:n_det], b[:n_det]
        if not torch.equal(a, b):
            d = (a.float() - b.float()).abs()
            bad.append((i, int((d > 0).sum()), float(d.max())))
    print(f'{k:24s}', 'identical' if not bad else f'DIFFERS {bad}', flush=True)
