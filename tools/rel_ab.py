"""Time the relation head kernel at the cfg3 shape (diagnostic; FGN_REL_WAVES = 4 / 8 / 16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
R, N, C = 300, 3, 1024
Q = torch.randn(R, 7, 7, C, generator=g).cuda()
S = torch.randn(N, 7, 7, C, generator=g).cuda()
rois = torch.zeros(R, 5).cuda()
gw, gb = torch.rand(C).cuda() + 0.5, torch.randn(C).cuda() * 0.1
fw, fb = torch.randn(6, C).cuda() * 0.1, torch.randn(6).cuda()
big = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
def run():
    return ops.relation_gn_head(Q, S, rois, gw, gb, fw, fb, N, 32, 1e-5)
for _ in range(3): run()
for cold in (False, True):
    tot = 0.0
    for _ in range(10):
        if cold: big.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print('waves', os.environ.get('FGN_REL_WAVES'), 'cold' if cold else 'warm', f'{tot / 10 * 1e3:.1f} us')
