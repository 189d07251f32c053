"""Time the F(4x4) transform kernels alone at the cfg3 layer shapes (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
g = torch.Generator().manual_seed(0)
def t(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('(the library no longer reads FGN_WG4_VEC / FGN_WG4_EAGER: this times the default form)')
for name, n, H, W, cin, cout, div in (('agrpn', 1, 50, 84, 1024, 1024, 3), ('sh300', 300, 7, 7, 512, 512, 1), ('sh100', 100, 7, 7, 512, 512, 1),
                                      ('mask0', 100, 7, 7, 1024, 256, 1), ('mask1', 100, 7, 7, 256, 256, 1), ('l3', 1, 50, 84, 256, 256, 1),
                                      ('l2', 1, 100, 167, 128, 128, 1), ('l1', 1, 200, 334, 64, 64, 1), ('spp_l3', 9, 16, 16, 256, 256, 1), ('spp_l2', 9, 32, 32, 128, 128, 1)):
    tiles = ((H + 3) // 4) * ((W + 3) // 4)
    t_pad = L.fgn_winograd_t_pad(n * div * tiles)
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    V = torch.empty(36, t_pad, cin, device='cuda')
    Mo = torch.randn(36, t_pad, cout, generator=g).cuda()
    y = torch.empty(n * div, H, W, cout, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    ti = t(lambda: L.fgn_winograd4_input_f32(x.data_ptr(), None, V.data_ptr(), None, n * div, div, H, W, cin, t_pad, st))
    to = t(lambda: L.fgn_winograd4_output_f32(Mo.data_ptr(), y.data_ptr(), None, None, n * div, H, W, cout, t_pad, 1, st))
    bi = (x.numel() * div + V.numel()) * 4 / 1e6
    bo = (Mo.numel() + y.numel()) * 4 / 1e6
    print(f'{name:7s} input {ti:6.1f} us ({bi / ti:6.2f} TB/s)  output {to:6.1f} us ({bo / to:6.2f} TB/s)', flush=True)
