"""Time GEMM-shaped launches of the conv kernels at cfg3 shapes with HIP events (diagnostic; env knobs of
csrc/conv_igemm.hip apply: FGN_PW_PERSIST, FGN_PW_M16, ...).  usage: gemm_time.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
g = torch.Generator().manual_seed(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, (n, tiles, cin, cout) in {'wino agrpn': (3, 273, 1024, 1024), 'wino sh300': (300, 4, 512, 512),
                                    'wino sh100': (100, 4, 512, 512), 'wino mask0': (100, 4, 1024, 256),
                                    'wino l3': (1, 273, 256, 256), 'wino l2': (1, 1050, 128, 128)}.items():
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(36, t_pad, cin, generator=g).cuda()
    U = (torch.randn(36, (cout + 127) // 128 * 128, cin, generator=g) * 0.03).cuda()
    Mo = torch.empty(36, t_pad, cout, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None, n, tiles, t_pad, cin, cout,
                                                U.shape[1], 36, None, None, 0, st))
    flop = 2.0 * 36 * n * tiles * cin * cout
    print(f'{name:14s} {us:8.1f} us  {flop / us / 1e6:6.1f} TF/s', flush=True)
for name, (rows, cin, cout, res) in {'relq 14700x1024>1024': (14700, 1024, 1024, False), 'sh conv3 14700x512>1024': (14700, 512, 1024, True),
                                     'sh conv1 14700x1024>512': (14700, 1024, 512, False), 'sh100 conv1 4900x1024>512': (4900, 1024, 512, False),
                                     'l3 conv1 4200x1024>256': (4200, 1024, 256, False), 'l3 conv3 4200x256>1024': (4200, 256, 1024, True),
                                     'l2 conv1 16700x512>128': (16700, 512, 128, False), 'l2 conv3 16700x128>512': (16700, 128, 512, True),
                                     'spp l3 conv1 2304x1024>256': (2304, 1024, 256, False)}.items():
    x = torch.randn(1, rows, 1, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.empty(1, rows, 1, cout, device='cuda')
    r = torch.randn(1, rows, 1, cout, generator=g).cuda() if res else None
    us = timeit(lambda: ops.conv2d(x, layer, residual=r, out=out))
    print(f'{name:28s} {us:8.1f} us  {2.0 * rows * cin * cout / us / 1e6:6.1f} TF/s', flush=True)
