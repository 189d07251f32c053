"""Saturated rate of the point-wise GEMM kernels on shapes WITHOUT a partial last round (tiles a whole multiple of the
resident workgroups for every tile shape): 16384 x K -> 1024.  usage: gemm_perfect.py [variants]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
variants = [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 2, 5, 7]
g = torch.Generator().manual_seed(0)
sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda')
ops._sched = lambda dev: sched
for rows, cin, cout in [(16384, 1024, 1024), (16384, 512, 1024), (32768, 1024, 512), (49152, 1024, 1024)]:
    x = torch.randn(1, rows, 1, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.zeros(1, rows, 1, cout, device='cuda')
    fn = lambda: ops.conv2d(x, layer, out=out, tile_hint=0)
    times = {v: [] for v in variants}
    for r in range(6):
        for v in variants:
            L.fgn_conv2d_tune(0, v)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            if r:
                times[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    L.fgn_conv2d_tune(0, -1)
    flop = 2.0 * rows * cin * cout
    print(f'{rows}x{cin}>{cout}: ' + ' | '.join(f'v{v} {statistics.median(times[v]):7.1f}us {flop / statistics.median(times[v]) / 1e6 / 157.3:.3f}' for v in variants), flush=True)
