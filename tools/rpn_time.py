import sys, torch, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fgn_amd import ops
from fgn_amd.config import fgn_r50_c4_config
g = torch.Generator().manual_seed(0)
fh, fw = 50, 84
cfg = fgn_r50_c4_config(3, 3); rp = cfg['rpn_head']
scores = torch.sigmoid(torch.randn(1, fh * fw * 15, generator=g) * 2.4).cuda()
deltas = (torch.randn(1, fh * fw * 15, 4, generator=g) * 0.35).cuda()
anchors = torch.from_numpy(ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], 16)).cuda()
for it in range(3):
    props, n, dbg = ops.rpn_proposals(scores, deltas, anchors, fh, fw, 16, 800, 1333, rp['target_means'],
                                      rp['target_stds'], 6000, 0, 0.7, 300, debug_topk=True)
torch.cuda.synchronize()
st = dbg[0, 8192 - 16: 8192 - 16 + 7].cpu().numpy().astype(np.int64)
names = ['load scores', 'radix select', 'compact', 'bitonic sort', 'decode+scan', 'nms', ]
print('n_props', int(n.item()))
for i, nm in enumerate(['load', 'select', 'compact', 'sort', 'decode', 'nms+out']):
    print(f'{nm:10s} {(st[i + 1] - st[i]) / 100.0:8.1f} us')
