"""Host-side cost of ops.gemm_tn / im2col calls at the training step's shapes (are the launches blocking?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
for (R, M, N) in ((6272, 1024, 512), (6272, 512, 4608), (6272, 512, 1024), (441, 512, 4608), (200, 1024, 9216), (6272, 1024, 1024), (2352, 256, 9216)):
    a = torch.randn(R, M, generator=g).cuda(); b = torch.randn(R, N, generator=g).cuda()
    for _ in range(3): ops.gemm_tn(a, b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): ops.gemm_tn(a, b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'R{R} M{M} N{N}: host {(t1 - t0) / 20 * 1e3:.3f} ms per call, incl. GPU drain {(t2 - t0) / 20 * 1e3:.3f} ms', flush=True)
