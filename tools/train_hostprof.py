"""cProfile of forward_train / Trainer.step on cfg3-size episodes: where the host time of the training path goes.
usage (GPU box): python tools/train_hostprof.py"""
import cProfile, os, pstats, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.train import Trainer
from fgn_amd.weights import init_state_dict
cfg = fgn_r50_c4_config(3, 3)
m = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
bs = [make_batch(i, 1, **CONFIGS['cfg3']) for i in range(3)]
m.forward_train(**bs[0]); m.forward_train(**bs[1])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(6):
    m.forward_train(**bs[i % 3])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
tr = Trainer(m)
tr.step(bs[0]); tr.step(bs[1])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(4):
    tr.step(bs[i % 3])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
