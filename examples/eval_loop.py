#!/usr/bin/env python3
"""The reference's evaluation loop (OptEvalHook._do_evaluate, main.py:269-326) on the MI355X path:
DataLoader(ds, batch_size=ds.batch, collate_fn=collate_fn_new) -> model.simple_test(**data, rescale=True)
-> chunked result pickles -> ds.evaluate(results_pkl_dir_fp=...).

    python examples/eval_loop.py --episodes 8 --batch 2 --height 320 --width 480
    python examples/eval_loop.py --dataset OMNIISEG --episodes 16 --n-ways 3 --k-shots 1     (cfg2-shaped)
"""
import argparse
import os
import sys
import tempfile
import time

import torch
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd.detector import FGN                          # noqa: E402
from fgn_amd.episodes import collate                      # noqa: E402
from fgn_amd.fewshot_ds import ClutteredCharsFewShotISEG, SyntheticFewShotISEG, write_chunked   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--episodes', type=int, default=8)
    ap.add_argument('--batch', type=int, default=4)       # eval_ds_cfg batch of the reference
    ap.add_argument('--n-ways', type=int, default=3)
    ap.add_argument('--k-shots', type=int, default=3)
    ap.add_argument('--height', type=int, default=800)
    ap.add_argument('--width', type=int, default=1333)
    ap.add_argument('--dataset', default='SYNTH', choices=['SYNTH', 'MNISTISEG', 'OMNIISEG'])
    ap.add_argument('--checkpoint', default=None, help='mmcv checkpoint of a trained reference FGN')
    args = ap.parse_args()

    if args.dataset == 'SYNTH':
        ds = SyntheticFewShotISEG(args.n_ways, args.k_shots, args.episodes, args.height, args.width, batch=args.batch)
    else:       # cluttered characters: 128^2 (MNISTISEG, cfg1) / 256^2 (OMNIISEG, cfg2) queries, 128^2 supports
        ds = ClutteredCharsFewShotISEG(args.dataset, args.n_ways, args.k_shots, n_imgs=args.episodes,
                                       img_size=128 if args.dataset == 'MNISTISEG' else 256, batch=args.batch)
    model = FGN(args.n_ways, args.k_shots)
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location='cpu'))
    loader = DataLoader(ds, batch_size=ds.batch, num_workers=2, collate_fn=collate)

    def results():
        for data in loader:
            yield model.simple_test(**data, rescale=True)

    with tempfile.TemporaryDirectory() as work_dir:
        out = os.path.join(work_dir, 'ResultsChunked')
        t0 = time.perf_counter()
        write_chunked(results(), out)
        dt = time.perf_counter() - t0
        metrics = ds.evaluate(results_pkl_dir_fp=out, model_dir=work_dir)
    print(f'{len(ds)} episodes in {dt:.2f} s ({len(ds) / dt:.1f} img/s incl. data loading and H2D)')
    print(metrics)


if __name__ == '__main__':
    main()
