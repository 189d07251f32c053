"""Time the grouped Winograd GEMM and the large 1x1 convs alone (diagnostic; A/B via env FGN_BAND_KB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
g = torch.Generator().manual_seed(0)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print('FGN_BAND_KB', os.environ.get('FGN_BAND_KB'))
G = 36
for name, n, tiles, cin, cout in (('agrpn', 3, 273, 1024, 1024), ('sh300', 300, 4, 512, 512), ('sh100', 100, 4, 512, 512),
                                  ('mask0', 100, 4, 1024, 256), ('mask1', 100, 4, 256, 256)):
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(G, t_pad, cin, generator=g).cuda()
    U = (torch.randn(G, (cout + 127) // 128 * 128, cin, generator=g) * 0.03).cuda()
    Mo = torch.empty(G, t_pad, cout, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.check(L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None, n, tiles, t_pad, cin, cout, U.shape[1], G, None, None, 0, st), 'g')
    ms = t(fn)
    fl = 2.0 * G * n * tiles * cin * cout
    print(f'wino gemm {name:6s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TF/s', flush=True)
for name, n, cin, cout, res in (('relQ 1024>1024 R300', 300, 1024, 1024, False), ('conv3 512>1024 R300', 300, 512, 1024, True),
                                ('conv1 1024>512 R300', 300, 1024, 512, False), ('conv3 R100', 100, 512, 1024, True),
                                ('conv1 R100', 100, 1024, 512, False), ('l3 1x1 256>1024', 0, 256, 1024, True),
                                ('l3 1x1 1024>256', 0, 1024, 256, False)):
    if n:
        x = torch.randn(n, 7, 7, cin, generator=g).cuda()
    else:
        x = torch.randn(1, 50, 84, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.empty(*x.shape[:3], cout, device='cuda')
    r = torch.randn(*out.shape, generator=g).cuda() if res else None
    ms = t(lambda: ops.conv2d(x, layer, residual=r, out=out))
    fl = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cin * cout
    print(f'conv {name:22s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.1f} TF/s', flush=True)
