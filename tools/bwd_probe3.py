import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch, torch.nn.functional as F
import test_hip_train as TT
from fgn_amd.config import tiny_config
from fgn_amd import train as TR
g = torch.Generator().manual_seed(1)
cfg = tiny_config(3, 2, width_div=2)
m, sd = TT._models(cfg)
C = cfg['roi_head']['shared_head']['inplanes']
x = torch.randn(40, C, 7, 7, generator=g).abs()
tr = TR.Trainer(m)
tape = []
TR.shared_head_train(m, x.permute(0, 2, 3, 1).contiguous().cuda(), 0.1, tape)
nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
xb = x
for b in range(3):
    p = f'roi_head.shared_head.{b}'
    bn = lambda t, i: F.batch_norm(t, None, None, sd[f'{p}.bn{i}.weight'], sd[f'{p}.bn{i}.bias'], True, 0.1, 1e-5)
    c1 = F.conv2d(xb, sd[p + '.conv1.weight']); p1 = bn(c1, 1); y1 = F.relu(p1)
    c2 = F.conv2d(y1, sd[p + '.conv2.weight'], padding=1); p2 = bn(c2, 2); y2 = F.relu(p2)
    c3 = F.conv2d(y2, sd[p + '.conv3.weight']); p3 = bn(c3, 3); out = F.relu(p3 + xb)
    t = tape[b]
    for nm, ref, pre in (('c1', c1, None), ('y1', y1, p1), ('c2', c2, None), ('y2', y2, p2), ('c3', c3, None), ('out', out, p3 + xb)):
        got = nchw(t[nm])
        msg = f'block {b} {nm}: max diff {float((got - ref).abs().max()):.2e} of range {float(ref.abs().max()):.2e}'
        if pre is not None:
            mm = ((got > 0) != (ref > 0))
            msg += f'  mask mismatches {int(mm.sum())} of {mm.numel()}  (|pre| at mismatches max {float(pre.abs()[mm].max()) if mm.any() else 0:.2e})'
            msg += f'  exact zeros in pre: {int((pre == 0).sum())}, frac positive {float((ref > 0).float().mean()):.3f}'
        print(msg)
    xb = out
