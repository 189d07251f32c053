"""Host-side time of each phase of a pipelined step (diagnostic): upload enqueue, kernel enqueue, download enqueue,
wait for the device, result packing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict

shape = CONFIGS['cfg3']
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
eps = []
for j in range(4):
    b = make_batch(j, 1, **shape)
    eps.append({k: (v.pin_memory() if isinstance(v, torch.Tensor) else [t.pin_memory() for t in v] if isinstance(v, list) else v)
                for k, v in b.items()})
T = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t
        return r
    return w
model._upload = timed('upload', model._upload)
model._detect_eager = timed('body', model._detect_eager)
model._start_download = timed('download', model._start_download)
pend = None
def step(i):
    global pend
    e = eps[i % 4]
    d = model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'], qry_isegmaps=e['qry_isegmaps'])
    if pend is not None:
        pe, pd = pend
        t = time.perf_counter(); pd[0]['host_ready'].synchronize(); T['wait'] = T.get('wait', 0.0) + time.perf_counter() - t
        t = time.perf_counter()
        model.pack_results(pd, 1, qry_bboxes=pe['qry_bboxes'], qry_cat_ids=pe['qry_cat_ids'], qry_isegmaps=pe['qry_isegmaps'], img_shape=pe['img_shape'], idx=pe['idx'])
        T['pack'] = T.get('pack', 0.0) + time.perf_counter() - t
    pend = (e, d)
for i in range(6):
    step(i)
torch.cuda.synchronize(); T.clear()
n = 30
t0 = time.perf_counter()
for i in range(n):
    step(i)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'ms/step {tot / n * 1e3:.2f}: ' + ', '.join(f'{k} {v / n * 1e3:.2f}' for k, v in T.items()))
