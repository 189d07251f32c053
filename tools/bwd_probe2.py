import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, torch.nn.functional as F
from fgn_amd import ops, train as TR
g = torch.Generator().manual_seed(1)
for (n, cin, cout) in ((40, 256, 256), (7, 128, 64), (40, 512, 128)):
    x = torch.randn(n, cin, 7, 7, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    w.requires_grad_(True)
    dy = torch.randn(n, cout, 7, 7, generator=g)
    y = F.conv2d(x, w, padding=1)
    (y * dy).sum().backward()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = TR._conv3x3_dgrad(dyd, w.detach().cuda()).permute(0, 3, 1, 2).cpu()
    dw = TR._conv3x3_wgrad(dyd, x.detach().permute(0, 2, 3, 1).contiguous().cuda()).cpu()
    e = (dx - x.grad).abs()
    print(n, cin, cout, 'dgrad rel', float(e.max() / x.grad.abs().max()), 'per-pixel max err', e.amax(dim=(0, 1)).numpy().round(5).tolist()[:2],
          'wgrad rel', float((dw - w.grad).abs().max() / w.grad.abs().max()))
# bn backward alone
P, C = 1960, 256
x = torch.randn(P, C, generator=g, requires_grad=True)
gam = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
bet = torch.zeros(C, requires_grad=True)
dy = torch.randn(P, C, generator=g)
y = F.relu(F.batch_norm(x.t()[None], None, None, gam, bet, True, 0.1, 1e-5)[0].t())
(y * dy).sum().backward()
yd, m, v = ops.bn_train(x.detach().cuda(), gam.detach().cuda(), bet.detach().cuda(), 1e-5, 0.1, relu=True, inplace=False)
dx, dg, db = ops.bn_train_backward(x.detach().cuda(), yd, dy.cuda(), m, v, gam.detach().cuda(), 1e-5)
print('bn bwd rel', float((dx.cpu() - x.grad).abs().max() / x.grad.abs().max()), float((dg.cpu() - gam.grad).abs().max() / gam.grad.abs().max()),
      float((db.cpu() - bet.grad).abs().max() / bet.grad.abs().max()))
