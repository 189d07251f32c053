"""The weight-gradient GEMM (fgn_gemm_tn_f32) against torch.matmul (rocBLAS) on the shapes of one training step:
error vs fp64 and time.  usage (GPU box): python tools/gemm_tn_bench.py"""
import os, sys, time
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
shapes = [(6272, 1024, 512), (6272, 512, 4608), (6272, 512, 1024), (441, 512, 4608), (200, 1024, 9216), (1568, 256, 9216),
          (1568, 256, 2304), (6272, 1024, 1024), (1000, 76, 1024), (37, 8, 12), (300, 1024, 4), (5000, 4, 256)]
for R, M, N in shapes:
    a = torch.randn(R, M, generator=g).cuda()
    b = torch.randn(R, N, generator=g).cuda()
    ref = (a.double().t() @ b.double())
    got = ops.gemm_tn(a, b)
    mm = a.t() @ b
    e1 = float((got.double() - ref).abs().max() / ref.abs().max())
    e2 = float((mm.double() - ref).abs().max() / ref.abs().max())
    def t(fn, n=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    u1, u2 = t(lambda: ops.gemm_tn(a, b)), t(lambda: a.t() @ b)
    fl = 2.0 * R * M * N
    print(f'R {R:6d} M {M:5d} N {N:5d}: err own {e1:.1e} rocBLAS {e2:.1e} | own {u1:8.1f} us {fl / u1 / 1e6:6.1f} TF/s | rocBLAS {u2:8.1f} us {fl / u2 / 1e6:6.1f} TF/s')
    assert e1 < 1e-5
