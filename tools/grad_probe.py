"""Gradient agreement of the HIP backward pass with torch.autograd on the training oracle, per tensor, with the Winograd
form of the 3x3 convolutions on and off (fewer ReLU sign flips between the two forward passes when off).
usage (GPU box): python tools/grad_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_hip_train as TT
from fgn_amd.config import tiny_config
from fgn_amd.episodes import make_batch
from fgn_amd.train import Trainer

cfg = tiny_config(3, 2, width_div=2)
b = make_batch(0, 2, 3, 2, 160, 224, 64)
m, sd = TT._models(cfg)
_, ref = TT._oracle_grads(sd, cfg, b, 5)
for wg in (4, 0):
    m, _ = TT._models(cfg)
    m.use_winograd = wg
    tr = Trainer(m)
    torch.manual_seed(5)
    tr.forward_backward(b)
    rep = TT._grad_report(tr.grads, ref)
    print('winograd', wg)
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f'  {k:50s} max/max {v[0]:.2e}  L2/L2 {v[1]:.2e}')
