#!/usr/bin/env python3
"""One GEMM shape on one instance of conv_pw_x3_kernel or conv_pw_h2_kernel, a few launches back to back (for rocprofv3
--pmc passes: tools/pmc_x3.sh).  usage: x3_one.py <shape substring of tools/x3_probe.py SHAPES> <bm> [launches] [x3 | h2]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import ops  # noqa: E402
from x3_probe import SHAPES  # noqa: E402

name, G, gr, valid, K, N = next(s for s in SHAPES if sys.argv[1] in s[0])
bm = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
g = torch.Generator().manual_seed(1)
x = torch.randn(G, gr, K, generator=g).relu_().cuda()
w = (torch.randn(G, N, K, generator=g) / K ** 0.5).cuda()
h2 = len(sys.argv) > 4 and sys.argv[4] == 'h2'
img = ops.pack_h2(w) if h2 else ops.pack_x3(w, mfma32=bm >= 2000)
out = torch.zeros(G, gr, N, device='cuda')
for _ in range(n):
    if h2:
        ops.gemm_h2(x, img, N, groups=G, grp_valid=valid, bm=bm, out=out)
    else:
        ops.gemm_x3(x, img, N, groups=G, grp_valid=valid, bm=bm, out=out)
torch.cuda.synchronize()
print('flop_per_launch', 2.0 * G * valid * K * N)
