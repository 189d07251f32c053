import sys; sys.path.insert(0,'/root/repo')
import torch
from fgn_amd.config import fgn_r50_c4_config, with_caps
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, RPN_MAX_PER_IMG, make_batch
from fgn_amd.weights import init_state_dict
for w, B in (('cfg3',1),('cfg4',4),('cfg5',1)):
    shape=CONFIGS[w]
    cfg=with_caps(fgn_r50_c4_config(shape['n_ways'],shape['k_shots']), rpn_max=RPN_MAX_PER_IMG.get(w))
    m=FGN(cfg['n_ways'],cfg['k_shots'],test_cfg=cfg['test_cfg'],state_dict=init_state_dict(cfg,0))
    b=make_batch(0,B,**shape)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(2): m.simple_test(**b,rescale=True)
    torch.cuda.synchronize()
    print(w,'batch',B,'peak allocated %.2f GB, reserved %.2f GB' % (torch.cuda.max_memory_allocated()/2**30, torch.cuda.max_memory_reserved()/2**30))
    del m; torch.cuda.empty_cache()
