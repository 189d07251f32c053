#!/usr/bin/env python3
"""conv_pw_h2_kernel (two-f16-plane products) and conv_pw_x3_kernel (three-bf16-plane products) against the f32-MFMA kernels: error against fp64 and time per launch on
the GEMM shapes of a cfg3 episode, the variants taking turns.  python tools/x3_probe.py [--reps 10]
FGN_HIP_LIB=tools/micro/libfgn_hip_exp.so adds the instances that were measured and not chosen (128-row tiles, 16x16x32
MFMA); FGN_HIP_LIB=tools/micro/libfgn_hip_x3ph.so --phases the phase clocks of one wave."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import ops  # noqa: E402

SHAPES = [   # name, groups, rows per group (allocated), valid rows, K, N
    ('relq 14700x1024>1024', 1, 14700, 14700, 1024, 1024),
    ('sh conv3 15141x512>1024', 1, 15141, 15141, 512, 1024),
    ('sh conv1 15141x1024>512', 1, 15141, 15141, 1024, 512),
    ('wino sh300 36x1236 512>512', 36, 1280, 1236, 512, 512),
    ('wino agrpn 36x819 1024>1024', 36, 896, 819, 1024, 1024),
    ('wino mask 36x400 512>512', 36, 512, 400, 512, 512),
    ('layer1 conv3 103664x64>256', 1, 103664, 103664, 64, 256),
    ('layer2 conv3 25916x128>512', 1, 25916, 25916, 128, 512),
    ('layer2 conv1 25916x512>128', 1, 25916, 25916, 512, 128),
    ('layer3 conv1 6504x1024>256', 1, 6504, 6504, 1024, 256),
    ('layer3 conv3 6504x256>1024', 1, 6504, 6504, 256, 1024),
]


def timed_round_robin(fns: dict, reps: int, rounds: int = 5) -> dict:
    """Median over `rounds` of the time per launch of every variant, the variants taking turns (the clock the chip holds
    depends on what ran just before: one variant after the other ranks them by their place in the queue)."""
    for fn in fns.values():
        fn()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1000.0 / reps)
    return {k: round(sorted(v)[len(v) // 2], 1) for k, v in res.items()}


# default instances (16x16x32 MFMA): 64 / 128 rows, nine terms; + 2000: the 32x32x16 form (experiments build)
VARIANTS = (('x6_bm64', 64, 6), ('x6_bm128', 128, 6), ('x9_bm64', 64, 9), ('x6_bm64_mfma32', 2064, 6), ('x6_bm128_mfma32', 2128, 6),
            ('x6_bm129_mfma32', 2129, 6))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--only', default='')
    ap.add_argument('--phases', action='store_true', help='phase clocks (FGN_HIP_LIB=tools/micro/libfgn_hip_x3ph.so)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(1)
    for name, G, gr, valid, K, N in SHAPES:
        if args.only and args.only not in name:
            continue
        x = torch.randn(G, gr, K, generator=g).relu_().to(dev)          # post-ReLU activations
        w = (torch.randn(G, N, K, generator=g) * (1.0 / K ** 0.5)).to(dev)
        shift = torch.randn(N, generator=g).to(dev)
        ref = torch.einsum('grk,gnk->grn', x[:, :valid].double(), w.double()) + shift.double()
        scale = ref.abs().max().item()
        img = ops.pack_x3(w)
        img32 = ops.pack_x3(w, mfma32=True)
        rec = dict(shape=name, gflop=2.0 * G * valid * K * N / 1e9)
        fns, outs = {}, {}
        # conv_pw_h2_kernel (three f16 products of scaled two-way splits): 64 / 128 rows
        imgh = ops.pack_h2(w)
        for tag, bm in (('h2_bm64', 64), ('h2_bm128', 128)):
            if G > 1 and gr % (128 if bm == 128 else 64):
                continue
            out = torch.zeros(G, gr, N, device=dev)
            fn = (lambda bm=bm, out=out: ops.gemm_h2(x, imgh, N, shift=shift, groups=G, grp_valid=valid, bm=bm, out=out))
            fn()
            torch.cuda.synchronize()
            d = (out[:, :valid].double() - ref).abs()
            outs[tag], fns[tag] = out, fn
            rec[tag] = dict(max_err=d.max().item() / scale, mean_err=d.mean().item() / scale)
        for tag, bm, nt in VARIANTS:
            if G > 1 and gr % (128 if bm % 1000 >= 128 else 64):
                continue
            out = torch.zeros(G, gr, N, device=dev)
            fn = (lambda bm=bm, nt=nt, out=out: ops.gemm_x3(x, img32 if bm >= 2000 else img, N, shift=shift, groups=G, grp_valid=valid,
                                                            bm=bm, nterms=nt, out=out))
            try:
                fn()
            except ops._lib.FgnHipError:        # an instance of the experiments build (FGN_HIP_LIB=tools/micro/libfgn_hip_exp.so)
                continue
            torch.cuda.synchronize()
            d = (out[:, :valid].double() - ref).abs()
            outs[tag], fns[tag] = out, fn
            rec[tag] = dict(max_err=d.max().item() / scale, mean_err=d.mean().item() / scale)
            if args.phases:
                import ctypes
                raw = ctypes.CDLL(ops._lib.LIB_PATH)
                buf = (ctypes.c_ulonglong * 16)()
                raw.fgn_x3_phases(buf)                      # clear
                fn()
                if raw.fgn_x3_phases(buf) == 0:
                    v = list(buf[0:7])
                    steps = max(v[5], 1)
                    rec[tag]['wg0_cycles_per_ktile'] = dict(
                        wait_barrier=round(v[0] / steps), issue=round(v[1] / steps), lds_landed=round(v[2] / steps),
                        split_mfma=round(v[3] / steps), epilogue_per_ktile=round(v[4] / steps), ktiles=v[5], kernel_cycles=v[6])
        # the f32 MFMA path: a 1x1 convolution packed for it (or the Winograd grouped GEMM entry) on the same operands
        if G == 1:
            with ops.gemm_math('f32'):
                layer = ops.pack_conv(w[0].reshape(N, K, 1, 1), bias=shift).to(dev)
            xin = x[0].reshape(1, gr, 1, K)
            out = torch.zeros(1, gr, 1, N, device=dev)
            fn = lambda: ops.conv2d(xin, layer, out=out)  # noqa: E731
            fn()
            torch.cuda.synchronize()
            d = (out.reshape(gr, N)[:valid].double() - ref[0]).abs()
        else:
            L = ops._lib.load()
            cout_pad = (N + 127) // 128 * 128
            u = torch.zeros(G, cout_pad, K, device=dev)
            u[:, :N] = w
            out = torch.zeros(G, gr, N, device=dev)
            fn = lambda: ops._lib.check(L.fgn_winograd_gemm_f32(x.data_ptr(), u.data_ptr(), out.data_ptr(), None, 1, valid, gr, K, N,  # noqa: E731
                                                                cout_pad, G, torch.cuda.current_stream().cuda_stream), 'wg')
            fn()
            torch.cuda.synchronize()
            d = (out[:, :valid].double() - (ref - shift.double())).abs()
        rec['f32_mfma'] = dict(max_err=d.max().item() / scale, mean_err=d.mean().item() / scale)
        fns['f32_mfma'] = fn
        for k, us in timed_round_robin(fns, args.reps).items():
            rec[k]['us'] = us
            rec[k]['tflops_f32_equiv'] = round(rec['gflop'] / us * 1e-3, 1)
        rec['row_tiles_equal'] = bool('x6_bm128' not in outs or torch.equal(outs['x6_bm64'][:, :valid], outs['x6_bm128'][:, :valid]))
        rec['auto_row_tile'] = ops._lib.load().fgn_x3_row_tile(G * gr, N, K, gr if G > 1 else 0, valid if G > 1 else 0)
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
