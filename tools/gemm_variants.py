"""Interleaved A/B of the point-wise GEMM kernels on the GEMM-shaped launches of a cfg3 episode (one process, rounds of
all variants per shape, medians: cdna_hip_programming.md 5.4 rule 24).  Variant 0 = the dispatcher's round-3 choice
(conv_pw_persist_kernel / conv_igemm_dma_kernel), 1..8 (+10 x stages) = conv_pw_persist2_kernel tile codes (fgn_conv2d_tune
knob 0), 1001..1003 = conv_pw_streamk_kernel in mode 1..3 (knob 2).
Every variant's output is compared with variant 0's.  usage: gemm_variants.py [rounds] [reps] [variants, e.g. 0,1,5]"""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd import ops, lib
L = lib.load()
g = torch.Generator().manual_seed(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0, 1, 2, 3, 4, 5]
SCHED = os.environ.get('FGN_GEMM_SCHED', '1') != '0'    # (ops.conv2d reads the same switch)
_SCH = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda') if SCHED else None
ops._sched = lambda dev: _SCH          # one workspace for all (serialised) launches of this tool: no fill kernel per call
_BASE = {0: 'r3', 1: '128x128', 2: '64x128', 3: '128x64', 4: '64x64', 5: '128x128w8', 6: '64x64w8', 7: '64x128w8', 8: '32x64'}
NAMES = {v: f'sk{v - 1000}' if v >= 1000 else _BASE[v % 10] + (f's{v // 10}' if v >= 10 else '') for v in variants}
_WS = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')       # Stream-K pieces (>= fgn_winograd_gemm_workspace_bytes)


def set_variant(v):
    L.fgn_conv2d_tune(0, 0 if v >= 1000 else v)
    L.fgn_conv2d_tune(2, (v - 1000) % 10 if v >= 1000 else 0)
    L.fgn_conv2d_tune(5, (v - 1000) // 10 if v >= 1000 else 0)      # 1013: pieces dropped (timing only)


def time_once(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run_shape(name, fn, out, flop):
    res, ref = {}, None
    for v in variants:
        set_variant(v)
        out.zero_()
        fn(); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
            res[v] = dict(err=0.0)
        else:
            res[v] = dict(err=float((out - ref).abs().max()))
        fn(); fn()
    times = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            set_variant(v)
            times[v].append(time_once(fn))
    L.fgn_conv2d_tune(0, -1)
    L.fgn_conv2d_tune(2, 0)
    line = f'{name:30s}'
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        res[v].update(us=round(med, 1), us_min=round(mn, 1), tflops=round(flop / med / 1e6, 1))
        line += f' | {NAMES[v]} {med:7.1f}us {flop / med / 1e6:6.1f}TF' + (f' e{res[v]["err"]:.0e}' if res[v]['err'] else '')
    print(line, flush=True)
    return res


table = {}
for name, (n, tiles, cin, cout) in {'wino agrpn 3x273 1024>1024': (3, 273, 1024, 1024), 'wino sh300 300x4 512>512': (300, 4, 512, 512),
                                    'wino sh100 100x4 512>512': (100, 4, 512, 512), 'wino mask0 100x4 1024>256': (100, 4, 1024, 256),
                                    'wino mask1 100x4 256>256': (100, 4, 256, 256),
                                    'wino l3 q+s 273+9x16 256>256': (1, 273 + 144, 256, 256), 'wino l2 q+s 128>128': (1, 1050 + 9 * 64, 128, 128),
                                    'wino l1 q+s 64>64': (1, 4200 + 9 * 256, 64, 64)}.items():
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(36, t_pad, cin, generator=g).cuda()
    V[:, n * tiles:] = 0
    U = (torch.randn(36, (cout + 127) // 128 * 128, cin, generator=g) * 0.03).cuda()
    Mo = torch.zeros(36, t_pad, cout, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda') if SCHED else None
    fn = lambda: L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None, n, tiles, t_pad, cin, cout,
                                         U.shape[1], 36, None if sched is None else sched.data_ptr(), _WS.data_ptr(), _WS.numel(), st)
    table[name] = run_shape(name, fn, Mo, 2.0 * 36 * n * tiles * cin * cout)
    del V, U, Mo
for name, (rows, cin, cout, res) in {'relq 14700x1024>1024': (14700, 1024, 1024, False), 'sh conv3 14700x512>1024 +res': (14700, 512, 1024, True),
                                     'sh conv1 14700x1024>512': (14700, 1024, 512, False), 'sh100 conv3 4900x512>1024 +res': (4900, 512, 1024, True),
                                     'sh100 conv1 4900x1024>512': (4900, 1024, 512, False),
                                     'l3 conv1 q+s 6504x1024>256': (4200 + 2304, 1024, 256, False), 'l3 conv3 q+s 6504x256>1024 +res': (6504, 256, 1024, True),
                                     'l2 conv1 q+s 26016x512>128': (16800 + 9216, 512, 128, False), 'l2 conv3 q+s 26016x128>512 +res': (26016, 128, 512, True),
                                     'l1 conv1 q+s 104064x256>64': (67200 + 36864, 256, 64, False), 'l1 conv3 q+s 104064x64>256 +res': (104064, 64, 256, True),
                                     'commute conv1 4200x1024>512': (4200, 1024, 512, False),
                                     'mask deconv 4900x256>1024': (4900, 256, 1024, False)}.items():
    x = torch.randn(1, rows, 1, cin, generator=g).cuda()
    layer = ops.pack_conv(torch.randn(cout, cin, 1, 1, generator=g) * 0.03, bias=torch.randn(cout, generator=g), relu=True).to('cuda')
    out = torch.zeros(1, rows, 1, cout, device='cuda')
    r = torch.randn(1, rows, 1, cout, generator=g).cuda() if res else None
    fn = lambda: ops.conv2d(x, layer, residual=r, out=out)
    table[name] = run_shape(name, fn, out, 2.0 * rows * cin * cout)
    del x, out, r
if os.environ.get('GEMM_VARIANTS_JSON'):
    json.dump(table, open(os.environ['GEMM_VARIANTS_JSON'], 'w'), indent=1)
