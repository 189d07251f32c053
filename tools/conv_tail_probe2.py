"""Where are the occupancy steps of the 64x64 DMA kernel? 1x1 conv 256->1024, M swept so that the
tile count crosses 1024 / 1280 (4 or 5 workgroups per CU x 256 CUs)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
cin, cout = 256, 1024
layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.05, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
for mt in (48, 60, 64, 65, 66, 72, 80, 81, 96, 120, 128, 129, 160):
    M = mt * 64
    x = torch.randn(1, M, 1, cin, generator=g).cuda()
    out = torch.empty(1, M, 1, cout, device='cuda')
    for _ in range(5):
        ops.conv2d(x, layer, out=out, tile_hint=4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        ops.conv2d(x, layer, out=out, tile_hint=4)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'm-tiles={mt:4d} blocks={mt * 16:5d}  {ms * 1e3:7.1f} us  {2.0 * M * cout * cin / ms / 1e9:6.1f} TF/s')
