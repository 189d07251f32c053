"""Time RoIAlign on the proposals / detections of a real cfg3 episode (two maps: C4 1024 ch + commuted conv1 512 ch).
FGN_HIP_LIB selects an alternative library build for an A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fgn_amd import ops
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict
shape = CONFIGS['cfg3']
cfg = fgn_r50_c4_config(3, 3)
model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
b = make_batch(0, 1, **shape)
e = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
e['img_shape'] = e['img_shape'].cpu()
model.debug_trace = tr = {}
model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'])
torch.cuda.synchronize()
model.debug_trace = None
fmap = tr['qry_fmap']
g_map = ops.conv2d(fmap, model._P['sh0_lin'])
pi = tr['per_image'][0]
for name, rois in (('300 proposals', pi['rois'].contiguous()),
                   ('100 detections', torch.cat([torch.zeros(100, 1, device='cuda'), pi['det'][:, :4]], 1).contiguous())):
    wh = (rois[:, 3] - rois[:, 1]).mean().item(), (rois[:, 4] - rois[:, 2]).mean().item()
    f = lambda: ops.roi_align2(fmap, g_map, rois, 7, 1 / 16, 0, True, None, post_shift2=model._P['sh0_shift'], relu2=True)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: mean box {wh[0]:.0f} x {wh[1]:.0f} px, roi_align2 {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call')
