"""Split-K plan of the mid-size point-wise layers: time per forced split count (FGN_CONV_SPLITS) - diagnostic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
SH = {'l3 conv1 4200x1024>256': (4200, 1024, 256), 'spp l3 conv1 2304x1024>256': (2304, 1024, 256), 'spp l3 conv3 2304x256>1024': (2304, 256, 1024),
      'l2 conv1 16700x512>128': (16700, 512, 128), 'spp l2 conv1 9216x512>128': (9216, 512, 128), 'spp l2 conv3 9216x128>512': (9216, 128, 512),
      'sppsh conv1 441x1024>512': (441, 1024, 512), 'sppsh conv3 441x512>1024': (441, 512, 1024), 'mask up 4900x256>1024': (4900, 256, 1024),
      'rpn head 12600x1024>76': (12600, 1024, 76), 'l3 down 4200x512>1024': (4200, 512, 1024)}
for name, (rows, cin, cout) in SH.items():
    x = torch.randn(1, rows, 1, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.empty(1, rows, 1, cout, device='cuda')
    fn = lambda: ops.conv2d(x, layer, out=out)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print(f'splits {os.environ.get("FGN_CONV_SPLITS", "auto"):>4s}  {name:30s} {us:7.1f} us {2.0 * rows * cin * cout / us / 1e6:6.1f} TF/s', flush=True)
