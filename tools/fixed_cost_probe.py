"""Fixed cost of one conv launch: 1024 workgroups of the 64x64 kernel (4 per CU), K-tiles swept (diagnostic).
time(KT) ~ fixed + KT * per_tile; also with 256 / 2048 / 4096 workgroups."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
NOUT = int(sys.argv[1]) if len(sys.argv) > 1 else 512     # Cout: A rows are reused by NOUT/64 column tiles (L2)
for blocks in (256, 1024, 2048, 4096):
    M = blocks * 64 // (NOUT // 64)
    line = f'blocks {blocks:5d} (M={M}, N={NOUT}): '
    for kt in (1, 2, 4, 8, 16, 32, 64):
        cin = 32 * kt
        x = torch.randn(M, 1, 1, cin, generator=g).cuda()
        layer = ops.pack_conv(torch.randn(NOUT, cin, 1, 1, generator=g) * 0.05, bias=torch.randn(NOUT, generator=g), relu=True).to('cuda')
        out = torch.empty(M, 1, 1, NOUT, device='cuda')
        for _ in range(5):
            ops.conv2d(x, layer, out=out, tile_hint=-4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            ops.conv2d(x, layer, out=out, tile_hint=-4)
        e1.record(); torch.cuda.synchronize()
        line += f'KT{kt:2d} {e0.elapsed_time(e1) / 50 * 1e3:6.1f}us  '
    print(line, flush=True)
