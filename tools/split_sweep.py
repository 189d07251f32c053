"""Sweep forced split-K counts over the mid-size cfg3 layers (diagnostic; one process per setting)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from fgn_amd import ops
SH = [('l3 1x1 1024>256', 1,50,84,1024,256,1,1,False), ('l3 3x3 256', 1,50,84,256,256,3,1,False), ('l3 1x1 256>1024', 1,50,84,256,1024,1,1,True),
      ('sh100 1x1 1024>512', 100,7,7,1024,512,1,1,False), ('sh100 1x1 512>1024', 100,7,7,512,1024,1,1,True),
      ('mask 3x3 256', 100,7,7,256,256,3,1,False), ('mask up 256>1024', 100,7,7,256,1024,1,1,False),
      ('l2 3x3 128', 1,100,167,128,128,3,1,False), ('l2 1x1 512>128', 1,100,167,512,128,1,1,False),
      ('spp l3 3x3 256', 9,16,16,256,256,3,1,False), ('spp l3 1x1 1024>256', 9,16,16,1024,256,1,1,False),
      ('spp l2 3x3 128', 9,32,32,128,128,3,1,False), ('spp sh 3x3', 9,7,7,512,512,3,1,False)]
g = torch.Generator().manual_seed(0)
out = []
for name, n, H, W, cin, cout, k, s, res in SH:
    x = torch.randn(n, H, W, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.05
    layer = ops.pack_conv(wt, bias=torch.randn(cout, generator=g), stride=s, pad=k // 2, relu=True).to('cuda')
    y = torch.empty(n, H, W, cout, device='cuda')
    r = torch.randn(n, H, W, cout, generator=g).cuda() if res else None
    for _ in range(3): ops.conv2d(x, layer, residual=r, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): ops.conv2d(x, layer, residual=r, out=y)
    e1.record(); torch.cuda.synchronize()
    out.append('%%6.1f' %% (e0.elapsed_time(e1) / 30 * 1e3))
print(' '.join(out))
''' % ROOT
print('columns: l3-1x1a l3-3x3 l3-1x1b sh100a sh100b mask3x3 maskup l2-3x3 l2-1x1 spp-l3-3x3 spp-l3-1x1 spp-l2-3x3 spp-sh3x3  (us)')
for s in (os.environ.get('SWEEP', 'auto,1,2,3,4,5,6,8,12')).split(','):
    env = dict(os.environ)
    if s != 'auto':
        env['FGN_CONV_SPLITS'] = s
    r = subprocess.run([sys.executable, '-c', CODE], env=env, capture_output=True, text=True)
    print(f'splits={s:5s}', r.stdout.strip().split('\n')[-1] if r.returncode == 0 else r.stderr[-300:], flush=True)
