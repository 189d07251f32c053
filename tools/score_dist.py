import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
cfg = fgn_r50_c4_config(3,3)
m = FGN(3,3,state_dict=init_state_dict(cfg,0)); m.debug_trace={}
b = make_batch(0,1,**CONFIGS['cfg3'])
m.simple_test(**b, rescale=True)
s = m.debug_trace['rpn_scores'][0].cpu().numpy()
srt = np.sort(s)[::-1]
print('top scores', srt[:5], 'rank1536', srt[1535], 'rank2048', srt[2047], 'rank6000', srt[5999], 'n==1.0', (s==1.0).sum(), 'n>0.999', (s>0.999).sum())
u = s.view(np.uint32).astype(np.uint64)
ordered = np.where(u & 0x80000000, ~u & 0xffffffff, u | 0x80000000).astype(np.uint64)
h = (~ordered) & 0xffffffff
bins16 = (h >> 16).astype(np.int64)
cnt = np.bincount(bins16, minlength=65536); cum = np.cumsum(cnt)
bstar = int(np.searchsorted(cum, 1536)); print('16-bit: b*', bstar, 'cum', cum[bstar], 'in bin', cnt[bstar])
bins22 = (h >> 10).astype(np.int64); c22 = np.bincount(bins22); cum22=np.cumsum(c22); b22=int(np.searchsorted(cum22,1536)); print('22-bit: cum', cum22[b22], 'in bin', c22[b22])
