"""Repeat each cfg3 conv shape N times and compare bitwise (diagnostic for data races)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
SH = [('stem', 1, 800, 1333, 4, 64, 7, 2, False), ('spp stem', 9, 256, 256, 4, 64, 7, 2, False),
      ('l1 1x1 64>64', 1, 200, 334, 64, 64, 1, 1, False), ('l1 3x3', 1, 200, 334, 64, 64, 3, 1, False),
      ('l1 1x1 64>256 res', 1, 200, 334, 64, 256, 1, 1, True), ('l1 1x1 256>64', 1, 200, 334, 256, 64, 1, 1, False),
      ('l2 3x3 s2', 1, 200, 334, 128, 128, 3, 2, False), ('l2 1x1 res', 1, 100, 167, 128, 512, 1, 1, True),
      ('l2 3x3', 1, 100, 167, 128, 128, 3, 1, False), ('l2 ds s2', 1, 200, 334, 256, 512, 1, 2, False),
      ('l3 3x3', 1, 50, 84, 256, 256, 3, 1, False), ('l3 1x1 res', 1, 50, 84, 256, 1024, 1, 1, True),
      ('l3 1x1', 1, 50, 84, 1024, 256, 1, 1, False), ('sh300 1x1', 300, 7, 7, 1024, 512, 1, 1, False)]
g = torch.Generator().manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, n, H, W, cin, cout, k, s, res in SH:
    pad = k // 2
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    wt = torch.randn(cout, cin if cin != 4 else 3, k, k, generator=g) * 0.05
    layer = ops.pack_conv(wt, bias=torch.randn(cout, generator=g), stride=s, pad=pad, relu=True,
                          pad_cin_to=4 if cin == 4 else None).to('cuda')
    ho, wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    r = torch.randn(n, ho, wo, cout, generator=g).cuda() if res else None
    ref = ops.conv2d(x, layer, residual=r).clone()
    bad = 0
    for i in range(N):
        y = ops.conv2d(x, layer, residual=r)
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            idx = torch.nonzero(d.reshape(-1, cout) > 0)
            print(f'   run {i}: {int((d > 0).sum())} elems differ, max {float(d.max()):.3g}, rows {idx[:, 0].min().item()}..{idx[:, 0].max().item()} cols {idx[:, 1].min().item()}..{idx[:, 1].max().item()}')
    print(f'{name:20s} {"OK" if bad == 0 else f"{bad}/{N} RUNS DIFFER"}', flush=True)

# Winograd layers (transforms + grouped GEMM) at the cfg3 sizes
for name, n, H, W, cin, cout, div in [('wg agrpn', 1, 50, 84, 1024, 1024, 3), ('wg sh300', 300, 7, 7, 512, 512, 1),
                                      ('wg sh100', 100, 7, 7, 512, 512, 1), ('wg l3', 1, 50, 84, 256, 256, 1),
                                      ('wg mask0', 100, 7, 7, 1024, 256, 1)]:
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.02
    layer = ops.pack_winograd(wt, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    s = (torch.rand(n * div, cin, generator=g) + 0.5).cuda() if (div > 1 or 'mask0' in name) else None
    ref = ops.conv3x3_winograd(x, layer, in_scale=s, a_img_div=div).clone()
    bad = sum(0 if torch.equal(ops.conv3x3_winograd(x, layer, in_scale=s, a_img_div=div), ref) else 1 for _ in range(N))
    print(f'{name:20s} {"OK" if bad == 0 else f"{bad}/{N} RUNS DIFFER"}', flush=True)
