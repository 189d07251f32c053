"""Where the shared-head backward starts to deviate from autograd (identical inputs), and the precision of the rocBLAS
fp32 GEMM the weight / data gradients go through.  usage (GPU box): python tools/bwd_probe.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import test_hip_train as TT
from fgn_amd.config import tiny_config
from fgn_amd import train as TR
from oracle import fgn_train_cpu as T

g = torch.Generator().manual_seed(1)
a, b = torch.randn(2000, 512, generator=g), torch.randn(512, 256, generator=g)
ref = (a.double() @ b.double())
for name, fn in (('torch.matmul cuda', lambda: (a.cuda() @ b.cuda()).cpu()), ('torch.matmul cpu', lambda: a @ b)):
    out = fn()
    print(name, 'rel err', float((out.double() - ref).abs().max() / ref.abs().max()))
print('allow_tf32', torch.backends.cuda.matmul.allow_tf32, 'precision', torch.get_float32_matmul_precision())

cfg = tiny_config(3, 2, width_div=2)
m, sd = TT._models(cfg)
C = cfg['roi_head']['shared_head']['inplanes']
x = torch.randn(40, C, 7, 7, generator=g).abs()
dout = torch.randn(40, C, 7, 7, generator=g)
ref_sd = {k: v.clone() for k, v in sd.items()}
names = [k for k in ref_sd if k.startswith('roi_head.shared_head') and 'running_' not in k]
for k in names:
    ref_sd[k].requires_grad_(True)
out = T.shared_head_train(x, ref_sd, cfg)
(out * dout).sum().backward()
tr = TR.Trainer(m)
tape, grads = [], {}
got = TR.shared_head_train(m, x.permute(0, 2, 3, 1).contiguous().cuda(), 0.1, tape)
TR._shared_backward(m, tr.W, tape, dout.permute(0, 2, 3, 1).contiguous().cuda(), grads)
rep = TT._grad_report(grads, {k: ref_sd[k].grad for k in names})
for k in names:
    print(f'{k:45s} max/max {rep[k][0]:.2e} L2 {rep[k][1]:.2e}')
