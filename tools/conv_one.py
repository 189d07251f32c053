"""Run one conv shape/tile repeatedly (for rocprofv3 --pmc). usage: shape_letter tile reps"""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fgn_amd import ops
letter, tile, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
SH = {'A': (3, 50, 84, 1024, 1024, 3, 1, 3, False), 'B': (300, 7, 7, 512, 512, 3, 1, 0, False),
      'D': (1, 200, 334, 64, 256, 1, 1, 0, True), 'C': (300, 7, 7, 1024, 512, 1, 1, 0, False),
      'F': (1, 50, 84, 256, 256, 3, 1, 0, False), 'G': (1, 50, 84, 256, 1024, 1, 1, 0, True),
      'L': (1, 100, 167, 128, 128, 3, 1, 0, False), 'I': (100, 7, 7, 512, 512, 3, 1, 0, False)}
n, H, W, cin, cout, k, s, div, res = SH[letter]
g = torch.Generator().manual_seed(0)
pad = k // 2
x = torch.randn(n // div if div else n, H, W, cin, generator=g).cuda()
layer = ops.pack_conv(torch.randn(cout, cin, k, k, generator=g) * 0.05, bias=torch.randn(cout, generator=g),
                      stride=s, pad=pad, relu=True).to('cuda')
ho, wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
out = torch.empty(n, ho, wo, cout, device='cuda')
r = torch.randn(n, ho, wo, cout, generator=g).cuda() if res else None
sc = (torch.rand(n, cin, generator=g) + 0.5).cuda() if div else None
for _ in range(reps):
    ops.conv2d(x, layer, residual=r, in_scale=sc, a_img_div=div if div else 1, out=out, tile_hint=tile)
torch.cuda.synchronize()
