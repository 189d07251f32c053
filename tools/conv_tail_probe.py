"""How much does grid-tail quantisation cost? AG-RPN-shaped conv at M giving 2.9 .. 4.1 waves of 1024 slots."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
cin = cout = 1024
wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.02
layer = ops.pack_conv(wt, bias=torch.randn(cout, generator=g), pad=1, relu=True).to('cuda')
for H, W in [(48, 64), (50, 84), (48, 85), (50, 82), (64, 64), (40, 64), (32, 64), (64, 68)]:
    n = 3
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    out = torch.empty(n, H, W, cout, device='cuda')
    M = n * H * W
    for tile in (4, 1):
        for _ in range(3):
            ops.conv2d(x, layer, out=out, tile_hint=tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            ops.conv2d(x, layer, out=out, tile_hint=tile)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        bm = 64 if tile == 4 else 128
        blocks = -(-M // bm) * (cout // bm)
        slots = 1024 if tile == 4 else 512
        print(f'M={M:6d} tile{tile} blocks={blocks:5d} waves={blocks / slots:5.2f}  {ms:7.3f} ms  {2.0 * M * cout * 9 * cin / ms / 1e9:6.1f} TF/s')
