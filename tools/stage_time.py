"""Stage timing of one cfg3 episode (diagnostic, not part of the product)."""
import sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
from fgn_amd import rle
shape = CONFIGS['cfg3']
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
b = make_batch(0, 1, **shape)
e = {k: (v.cuda() if isinstance(v, torch.Tensor) else [t.cuda() for t in v] if isinstance(v, list) else v) for k, v in b.items()}
e['img_shape'] = e['img_shape'].cpu()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dets = model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'])
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res = model.pack_results(dets, 1, img_shape=e["img_shape"])
    t3 = time.perf_counter()
    print(f'iter {it}: launch {1e3*(t1-t0):.1f} ms, gpu-done {1e3*(t2-t0):.1f} ms, pack {1e3*(t3-t2):.2f} ms, n={len(res[0]["dt_scores"])}')
