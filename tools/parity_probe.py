"""Measure (not assert) the end-to-end maxima against the oracle: python tools/parity_probe.py cfg1 cfg2 cfg3 [--no-wg]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd.agreement import episode_maxima  # noqa: E402

from fgn_amd.config import fgn_r50_c4_config, tiny_config  # noqa: E402
from fgn_amd.detector import FGN  # noqa: E402
from fgn_amd.episodes import CONFIGS, make_batch  # noqa: E402
from fgn_amd.weights import init_state_dict  # noqa: E402
from oracle import fgn_ref_cpu as O  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith('--')] or ['cfg1', 'cfg2']
wg = False if '--no-wg' in sys.argv else 2 if '--wg2' in sys.argv else True
for name in names:
    for seed in ((11, 12) if name != 'cfg3' else (21,)):
        if name == 'tiny':
            cfg = tiny_config(3, 2, width_div=2)
            batch = make_batch(seed, 1, 3, 2, 160, 224, 64)
        else:
            shape = CONFIGS[name]
            cfg = fgn_r50_c4_config(shape['n_ways'], shape['k_shots'])
            batch = make_batch(seed, 1, **shape)
        sd = init_state_dict(cfg, 0)
        model = FGN(cfg['n_ways'], cfg['k_shots'], backbone=cfg['backbone'], rpn_head=cfg['rpn_head'],
                    roi_head=cfg['roi_head'], test_cfg=cfg['test_cfg'], state_dict=sd)
        model.use_winograd = wg
        model.use_roi_commute = bool(wg)
        model.debug_trace = {}
        got = model.simple_test(**batch, rescale=True)
        tr = model.debug_trace
        t0 = time.time()
        tr_ref = {}
        ref = O.simple_test(sd, cfg, **batch, trace=tr_ref)
        n = int(tr['per_image'][0]['n_det'][0])
        m = episode_maxima(ref[0], got[0], tr_ref['mask_prob'].numpy(), tr['per_image'][0]['mask_prob'][:n].cpu().numpy(),
                           tr_ref['mask_logits'].numpy(), tr['per_image'][0]['mask_logits'][:n].cpu().numpy())
        r = tr_ref['qry_fmap']
        m['fmap_rel'] = float((tr['qry_fmap'].permute(0, 3, 1, 2).cpu() - r).abs().max() / r.abs().max())
        rf = tr_ref['roi_feats']
        m['cls_raw_absmax'] = float(tr_ref['cls_raw'].abs().max())
        m.update(cfg=name, seed=seed, winograd=wg, oracle_s=round(time.time() - t0, 1))
        print(json.dumps(m), flush=True)
