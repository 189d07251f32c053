import sys, numpy as np, torch
sys.path.insert(0, '.')
from fgn_amd.config import tiny_config, fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import make_batch, CONFIGS
from fgn_amd.weights import init_state_dict
for name, cfg, b in (('tiny', tiny_config(3, 2, width_div=2), make_batch(0, 2, 3, 2, 160, 224, 64)),
                     ('cfg3', fgn_r50_c4_config(3, 3), make_batch(1, 1, **CONFIGS['cfg3']))):
    m = FGN(cfg['n_ways'], cfg['k_shots'], backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
            test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    m.debug_trace = {}
    a = m.simple_test(**b, rescale=True); fa = m.debug_trace['qry_fmap'].clone(); sa = m.debug_trace['spp_fmaps'].clone()
    m.use_merged_backbone = True
    m.debug_trace = {}
    c = m.simple_test(**b, rescale=True); fb = m.debug_trace['qry_fmap']; sb = m.debug_trace['spp_fmaps']
    print(name, 'qry fmap max rel diff', float((fa - fb).abs().max() / fa.abs().max()), 'spp', float((sa - sb).abs().max() / sa.abs().max()),
          'dets', [len(x['dt_scores']) for x in a], [len(x['dt_scores']) for x in c],
          'max dscore', max(float(np.abs(x['dt_scores'][:10] - y['dt_scores'][:10]).max()) for x, y in zip(a, c)))
