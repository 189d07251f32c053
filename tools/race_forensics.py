"""When the sh300 1x1 conv produces a wrong 32x32 block, work out what the wrong values are (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops
g = torch.Generator().manual_seed(0)
n, H, W, cin, cout = 300, 7, 7, 1024, 512
x = torch.randn(n, H, W, cin, generator=g)
wt = torch.randn(cout, cin, 1, 1, generator=g) * 0.05
b = torch.randn(cout, generator=g)
layer = ops.pack_conv(wt, bias=b, relu=False).to('cuda')
xc = x.cuda()
X = x.reshape(-1, cin).double(); Wm = wt.reshape(cout, cin).double()
ref = ops.conv2d(xc, layer).clone()
for attempt in range(3):      # make sure the reference itself is clean: majority of 3
    r2 = ops.conv2d(xc, layer)
    if not torch.equal(r2, ref):
        ref = ops.conv2d(xc, layer).clone()
found = 0
for i in range(600):
    y = ops.conv2d(xc, layer)
    if torch.equal(y, ref):
        continue
    d = (y - ref).reshape(-1, cout)
    idx = torch.nonzero(d != 0)
    r0, r1, c0, c1 = idx[:, 0].min().item(), idx[:, 0].max().item(), idx[:, 1].min().item(), idx[:, 1].max().item()
    r0 = r0 // 32 * 32; c0 = c0 // 32 * 32
    bad = y.reshape(-1, cout)[r0:r0 + 32, c0:c0 + 32].double().cpu()
    good = ref.reshape(-1, cout)[r0:r0 + 32, c0:c0 + 32].double().cpu()
    rows = X[r0:r0 + 32]; cols = Wm[c0:c0 + 32]
    print(f'run {i}: block rows {r0}.. cols {c0}..  n_bad {(bad != good).sum().item()}')
    cands = {'full': rows @ cols.T + b[c0:c0 + 32].double()}
    for kcut in (8, 16, 24, 32, 64):
        cands[f'minus last {kcut} k'] = rows[:, :cin - kcut] @ cols[:, :cin - kcut].T + b[c0:c0 + 32].double()
        cands[f'minus first {kcut} k'] = rows[:, kcut:] @ cols[:, kcut:].T + b[c0:c0 + 32].double()
    for name, c in cands.items():
        e = (bad - c).abs().max().item()
        print(f'    vs {name:18s} max err {e:.3g}')
    # per-k-chunk attribution: which 8-wide k chunks are missing? solve bad - full = -sum_missing
    diff = bad - cands['full']
    best = []
    for k8 in range(0, cin, 4):
        part = rows[:, k8:k8 + 4] @ cols[:, k8:k8 + 4].T
        # correlation of diff with -part
        num = -(diff * part).sum().item(); den = (part * part).sum().item()
        best.append((num / den, k8))
    best.sort(reverse=True)
    print('    top k-chunk (4 wide) coefficients:', [(round(c, 2), k) for c, k in best[:10]])
    found += 1
    if found >= 3:
        break
print('done, found', found)
