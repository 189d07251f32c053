"""Run one GEMM-shaped launch of the dominant kernel N times (target of tools/pmc_gemm.sh).
usage: gemm_one.py <agrpn|sh300|sh100|relq|conv3> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
g = torch.Generator().manual_seed(0)
name = sys.argv[1] if len(sys.argv) > 1 else 'sh300'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
W = {'agrpn': (3, 273, 1024, 1024), 'sh300': (300, 4, 512, 512), 'sh100': (100, 4, 512, 512)}
if name in W:
    n, tiles, cin, cout = W[name]
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(36, t_pad, cin, generator=g).cuda()
    U = (torch.randn(36, (cout + 127) // 128 * 128, cin, generator=g) * 0.03).cuda()
    Mo = torch.empty(36, t_pad, cout, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    fn = lambda: L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None, n, tiles, t_pad, cin, cout, U.shape[1], 36, None, None, 0, st)
    flop = 2.0 * 36 * n * tiles * cin * cout
else:
    cin, cout, res = {'relq': (1024, 1024, False), 'conv3': (512, 1024, True)}[name]
    x = torch.randn(300, 7, 7, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.empty(300, 7, 7, cout, device='cuda')
    r = torch.randn(300, 7, 7, cout, generator=g).cuda() if res else None
    fn = lambda: ops.conv2d(x, layer, residual=r, out=out)
    flop = 2.0 * 300 * 49 * cin * cout
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print('flop_per_launch', flop)
