"""Wall time of the phases of Trainer.step (synchronised between phases) on cfg3-size episodes.
usage (GPU box): python tools/train_phases.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from fgn_amd import ops, train as TR
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
cfg = fgn_r50_c4_config(3, 3)
m = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
bs = [make_batch(i, 1, **CONFIGS['cfg3']) for i in range(3)]
tr = TR.Trainer(m)
tr.step(bs[0]); tr.step(bs[1])
acc = {}
def lap(name, t0):
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
n = 6
for i in range(n):
    b = bs[i % 3]
    torch.cuda.synchronize(); t = time.perf_counter()
    m._tape = {}
    losses = TR.forward_train(m, **b)
    t = lap('forward_train', t)
    tr.grads = TR.backward(m, tr.W, m._tape)
    m._tape = None
    t = lap('backward', t)
    for k, g in tr.grads.items():
        ops.adagrad_step(tr.W[k], g.contiguous(), tr.state[k], 0.005, 1e-5)
    t = lap('adagrad', t)
    heads = m._pack_heads(tr.W)
    t = lap('pack_heads', t)
    for k, v in heads.items():
        m._P[k] = v if not isinstance(v, torch.Tensor) else v.float().contiguous()
    TR.pack_train(m, tr.device, tr.W, tr.buffers)
    t = lap('pack_train', t)
print({k: round(v / n * 1e3, 2) for k, v in acc.items()})
