"""Winograd vs direct conv on the cfg3 3x3 layers (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
SHAPES = [('agrpn', 1, 50, 84, 1024, 1024, 3), ('sh300', 300, 7, 7, 512, 512, 1), ('sh100', 100, 7, 7, 512, 512, 1),
          ('mask0', 100, 7, 7, 1024, 256, 1), ('mask1', 100, 7, 7, 256, 256, 1), ('l3', 1, 50, 84, 256, 256, 1),
          ('l2', 1, 100, 167, 128, 128, 1), ('l1', 1, 200, 334, 64, 64, 1), ('spp_l3', 9, 16, 16, 256, 256, 1),
          ('spp_l2', 9, 32, 32, 128, 128, 1), ('spp_l1', 9, 64, 64, 64, 64, 1), ('spp_sh', 9, 7, 7, 512, 512, 1),
          ('mask2', 100, 7, 7, 256, 256, 1)]
g = torch.Generator().manual_seed(0)
for name, n, H, W, cin, cout, div in SHAPES:
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.02
    b = torch.randn(cout, generator=g)
    s = (torch.rand(n * div, cin, generator=g) + 0.5).cuda() if div > 1 else None
    wg = ops.pack_winograd(wt, bias=b, relu=True, m=2).to('cuda')
    wg4 = ops.pack_winograd(wt, bias=b, relu=True, m=4).to('cuda')
    dr = ops.pack_conv(wt, bias=b, pad=1, relu=True).to('cuda')
    flop = 2.0 * n * div * H * W * cout * 9 * cin

    def direct():
        xi = ops.scale_channels(x, s, div) if s is not None else x
        return ops.conv2d(xi, dr)

    def wino():
        return ops.conv3x3_winograd(x, wg, in_scale=s, a_img_div=div)

    def wino4():
        return ops.conv3x3_winograd(x, wg4, in_scale=s, a_img_div=div)
    res = []
    for fn in (direct, wino, wino4):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10)
    d = direct()
    err, err4 = (d - wino()).abs().max().item(), (d - wino4()).abs().max().item()
    print(f'{name:8s} direct {res[0]:.3f} ms ({flop/res[0]/1e9:6.1f} TF)  F(2x2) {res[1]:.3f} ms ({flop/res[1]/1e9:6.1f} TF eff) '
          f'F(4x4) {res[2]:.3f} ms ({flop/res[2]/1e9:6.1f} TF eff)  maxdiff {err:.2e} / {err4:.2e} of {d.abs().max().item():.1f}', flush=True)
