"""Host enqueue time vs GPU time of one cfg3 episode, eager launches vs hipGraph replay (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict

shape = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'cfg3']
cfg = fgn_r50_c4_config(shape['n_ways'], shape['k_shots'])
model = FGN(cfg['n_ways'], cfg['k_shots'], test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
dev = torch.device('cuda')
eps = []
for j in range(2):
    b = make_batch(j, 1, **shape)
    eps.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()})
    eps[-1]['img_shape'] = eps[-1]['img_shape'].cpu()
code = model.encode_supports(eps[0]['spp_imgs'], eps[0]['spp_bboxes'], eps[0]['spp_isegmaps'])
for cached in (False, True):
    for graphs in (False, True):
        model.use_graphs = graphs
        host, gpu = [], []
        for i in range(8):
            e = eps[i % 2]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            d = model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'],
                                    support_code=code if cached else None)
            e1.record()
            t1 = time.perf_counter()
            r = model.pack_results(d, 1)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            if i >= 3:
                host.append((t1 - t0) * 1e3); gpu.append(e0.elapsed_time(e1))
                pack = (t2 - t1) * 1e3
        print(f'cached={cached} graphs={graphs}: host enqueue {sum(host)/len(host):.2f} ms, '
              f'GPU span {sum(gpu)/len(gpu):.2f} ms, pack_results(after sync wait) {pack:.2f} ms', flush=True)
