"""Micro-benchmark of the conv kernel on the cfg3 layer shapes (diagnostic).
usage: python tools_conv_bench.py [tiles e.g. 0,1,2,3,4] [reps]"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fgn_amd import ops
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else '0,1,2,3,4').split(',')]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SHAPES = [
    # name, n_img, H, W, cin, cout, k, stride, in_scale(a_img_div), residual
    ('A agrpn3x3', 3, 50, 84, 1024, 1024, 3, 1, 3, False),
    ('B sh3x3 R300', 300, 7, 7, 512, 512, 3, 1, 0, False),
    ('C sh1x1 R300 1024>512', 300, 7, 7, 1024, 512, 1, 1, 0, False),
    ('C2 sh1x1 R300 512>1024', 300, 7, 7, 512, 1024, 1, 1, 0, True),
    ('D l1 1x1 64>256', 1, 200, 334, 64, 256, 1, 1, 0, True),
    ('Dn l1 1x1 64>256 nores', 1, 200, 334, 64, 256, 1, 1, 0, False),
    ('Jn l2 1x1 128>512 nores', 1, 100, 167, 128, 512, 1, 1, 0, False),
    ('D2 l1 3x3 64>64', 1, 200, 334, 64, 64, 3, 1, 0, False),
    ('E sh3x3 R9', 9, 7, 7, 512, 512, 3, 1, 0, False),
    ('F l3 3x3 256', 1, 50, 84, 256, 256, 3, 1, 0, False),
    ('G l3 1x1 256>1024', 1, 50, 84, 256, 1024, 1, 1, 0, True),
    ('H mask 3x3 1024>256', 100, 7, 7, 1024, 256, 3, 1, 0, False),
    ('I sh3x3 R100', 100, 7, 7, 512, 512, 3, 1, 0, False),
    ('J l2 1x1 128>512', 1, 100, 167, 128, 512, 1, 1, 0, True),
    ('K stem', 1, 800, 1333, 4, 64, 7, 2, 0, False),
]
g = torch.Generator().manual_seed(0)
print(f'{"shape":28s} ' + ' '.join(f'tile{t:>1d}: ms / TF  ' for t in tiles))
for name, n, H, W, cin, cout, k, s, div, res in SHAPES:
    pad = k // 2
    n_in = n // div if div else n
    x = torch.randn(n_in, H, W, cin, generator=g).cuda()
    wt = torch.randn(cout, cin if cin != 4 else 3, k, k, generator=g) * 0.05
    layer = ops.pack_conv(wt, bias=torch.randn(cout, generator=g), stride=s, pad=pad, relu=True,
                          pad_cin_to=4 if cin == 4 else None).to('cuda')
    ho, wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    out = torch.empty(n, ho, wo, cout, device='cuda')
    r = torch.randn(n, ho, wo, cout, generator=g).cuda() if res else None
    sc = (torch.rand(n, cin, generator=g) + 0.5).cuda() if div else None
    flop = 2.0 * n * ho * wo * cout * k * k * cin
    line = f'{name:28s} '
    for t in tiles:
        kw = dict(residual=r, in_scale=sc, a_img_div=div if div else 1, out=out, tile_hint=t)
        for _ in range(3):
            ops.conv2d(x, layer, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, layer, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        line += f'{ms:7.3f}/{flop / ms / 1e9:6.1f}  '
    print(line, flush=True)
