#!/usr/bin/env python3
"""The reference's training loop body (mmcv EpochBasedRunner + OptimizerHook + StepLrUpdaterHook with
fgn_train_schedule.py: Adagrad lr 0.005, weight decay 1e-5, lr_mult 0.1 under roi_head, step [3] x0.1, linear
warm-up 100 iterations, 3 epochs) on the MI355X path: DataLoader(ds, collate_fn_new) -> Trainer.step(batch)
-> mmcv-style checkpoint -> the trained heads evaluated through simple_test + ds.evaluate.

    python examples/train_loop.py --dataset OMNIISEG --episodes 32 --epochs 3
    python examples/train_loop.py --dataset SYNTH --height 320 --width 480 --episodes 16

The backbone is frozen as in fgn_r50_c4_densecl.py:31; with no pretrained checkpoint here it is randomly
initialised, so this demonstrates the loop (losses fall on the training episodes), not a trained detector.
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
from fgn_amd.fewshot_ds import ClutteredCharsFewShotISEG, SyntheticFewShotISEG   # noqa: E402
from fgn_amd.train import Trainer, step_lr               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--episodes', type=int, default=16)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--n-ways', type=int, default=3)
    ap.add_argument('--k-shots', type=int, default=1)
    ap.add_argument('--height', type=int, default=256)
    ap.add_argument('--width', type=int, default=256)
    ap.add_argument('--dataset', default='OMNIISEG', choices=['SYNTH', 'MNISTISEG', 'OMNIISEG'])
    ap.add_argument('--lr', type=float, default=0.005)
    ap.add_argument('--checkpoint', default=None, help='mmcv checkpoint to start from')
    ap.add_argument('--save', default=None)
    args = ap.parse_args()
    if args.dataset == 'SYNTH':
        ds = SyntheticFewShotISEG(args.n_ways, args.k_shots, args.episodes, args.height, args.width, batch=args.batch)
    else:
        ds = ClutteredCharsFewShotISEG(args.dataset, args.n_ways, args.k_shots, n_imgs=args.episodes,
                                       img_size=128 if args.dataset == 'MNISTISEG' else 256, batch=args.batch)
    model = FGN(args.n_ways, args.k_shots)
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location='cpu'))
    trainer = Trainer(model, lr=args.lr)
    it = 0
    for epoch in range(args.epochs):
        ds.reshuffle()
        loader = DataLoader(ds, batch_size=ds.batch, num_workers=2, collate_fn=collate)
        t0, tot, n = time.perf_counter(), {}, 0
        for data in loader:
            trainer.lr = step_lr(args.lr, it, epoch)
            losses = trainer.step(data)
            for k, v in losses.items():
                tot[k] = tot.get(k, 0.0) + float(v[0] if isinstance(v, list) else v)
            it, n = it + 1, n + 1
        dt = time.perf_counter() - t0
        print(f'epoch {epoch}: {n} iterations in {dt:.2f} s, lr {trainer.lr:.2e}, ' +
              ', '.join(f'{k} {v / n:.4f}' for k, v in tot.items()))
    sd = trainer.state_dict()
    if args.save:
        torch.save(trainer.checkpoint(meta={'iter': it, 'epoch': args.epochs}), args.save)   # resumable: Trainer.resume
    model.load_state_dict(sd)
    with tempfile.TemporaryDirectory() as work_dir:
        loader = DataLoader(ds, batch_size=ds.batch, num_workers=2, collate_fn=collate)
        results = [r for data in loader for r in model.simple_test(**data, rescale=True)]
        print(ds.evaluate(results=results, model_dir=work_dir))


if __name__ == '__main__':
    main()
