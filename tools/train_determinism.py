"""Is a training step bit-reproducible?  Two Trainers from the same weights, same episode, same seeds: gradients of the
first step and weights after a few steps compared bitwise, tensor by tensor.
usage (GPU box): python tools/train_determinism.py [--size cfg3|small]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.train import Trainer
from fgn_amd.weights import init_state_dict
small = '--size' in sys.argv and sys.argv[sys.argv.index('--size') + 1] == 'small'
cfg = fgn_r50_c4_config(3, 3)
sd = init_state_dict(cfg, 0)
b = make_batch(0, 1, 3, 3, 256, 320, 128) if small else make_batch(0, 1, **CONFIGS['cfg3'])
runs = []
for r in range(2):
    m = FGN(3, 3, state_dict=sd)
    t = Trainer(m)
    torch.manual_seed(0)
    t.forward_backward(b)
    g0 = {k: v.clone() for k, v in t.grads.items()}
    for it in range(3):
        torch.manual_seed(it)
        t.step(b)
    runs.append((g0, {k: v.clone() for k, v in t.W.items()}))
for name, i in (('gradients of step 0', 0), ('weights after 3 steps', 1)):
    diff = {k: float((runs[0][i][k] - runs[1][i][k]).abs().max()) for k in runs[0][i] if not torch.equal(runs[0][i][k], runs[1][i][k])}
    print(name, ': identical' if not diff else f': {len(diff)} of {len(runs[0][i])} tensors differ', dict(list(diff.items())[:8]))
