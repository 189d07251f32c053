"""Functional check of the training path: overfit the heads (frozen, randomly initialised backbone) on a handful of
cluttered-character episodes and watch the detector's AP50 on those same episodes through simple_test.
usage (GPU box): python tools/overfit_probe.py [--steps 300] [--episodes 2] [--lr 0.005]"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from fgn_amd.detector import FGN
from fgn_amd.episodes import collate
from fgn_amd.fewshot_ds import ClutteredCharsFewShotISEG
from fgn_amd.fsiseg_eval import evaluate_results
from fgn_amd.train import Trainer

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--episodes', type=int, default=2)
ap.add_argument('--lr', type=float, default=0.005)
ap.add_argument('--every', type=int, default=50)
ap.add_argument('--dataset', default='OMNIISEG', help='OMNIISEG / MNISTISEG (cluttered characters) or cfg1..cfg5 (bench episodes)')
a = ap.parse_args()
if a.dataset.startswith('cfg'):
    from fgn_amd.episodes import CONFIGS, make_batch
    batches = [make_batch(i, 1, **CONFIGS[a.dataset]) for i in range(a.episodes)]
    K = CONFIGS[a.dataset]['k_shots']
else:
    ds = ClutteredCharsFewShotISEG(a.dataset, 3, 1, n_imgs=max(a.episodes, 8), img_size=256, batch=a.episodes)
    batches = [collate([ds[i] for i in range(a.episodes)])]
    K = 1
m = FGN(3, K)
tr = Trainer(m, lr=a.lr)
t0 = time.perf_counter()
for it in range(a.steps + 1):
    if it % a.every == 0:
        m.load_state_dict(tr.state_dict())
        res = [r for b in batches for r in m.simple_test(**b, rescale=True)]
        ev = evaluate_results(res, 3)
        tr.refresh()                               # load_state_dict dropped the packed training layers
        print(f'step {it:4d} ({time.perf_counter() - t0:6.1f} s): detections {[len(r["dt_scores"]) for r in res]} '
              f'bbox AP50 {ev["bbox_mAP50"]:.3f} segm AP50 {ev["segm_mAP50"]:.3f}', flush=True)
    if it == a.steps:
        break
    torch.manual_seed(it)
    L = tr.step(batches[it % len(batches)])
    if it % a.every == 0:
        print('   losses', {k: round(float(v[0] if isinstance(v, list) else v), 4) for k, v in L.items()}, flush=True)
