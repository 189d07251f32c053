"""Phase times inside the two single-workgroup selection kernels (rpn_proposals_kernel, det_post_kernel) on the
tensors of a real cfg3 episode (debug trace of FGN._detect_body), from the kernels' own 100 MHz stamps, and the
wall time of each launch by HIP events (diagnostic)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import ops
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.weights import init_state_dict

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
shape = CONFIGS[which]
n, k = shape['n_ways'], shape['k_shots']
cfg = fgn_r50_c4_config(n, k)
if which == 'cfg5':
    cfg['test_cfg']['rpn']['max_per_img'] = 1000
model = FGN(n, k, state_dict=init_state_dict(cfg, 0), test_cfg=cfg['test_cfg'])
b = make_batch(0, 1, **shape)
e = {kk: (v.cuda() if isinstance(v, torch.Tensor) else v) for kk, v in b.items()}
e['img_shape'] = e['img_shape'].cpu()
model.debug_trace = tr = {}
model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'])
torch.cuda.synchronize()
model.debug_trace = None
rp, tc, rh = cfg['rpn_head'], cfg['test_cfg'], cfg['roi_head']
P = model._P
fh, fw = tr['qry_fmap'].shape[1:3]
ih, iw = int(e['img_shape'][0][0]), int(e['img_shape'][0][1])
scores, deltas = tr['rpn_scores'].contiguous(), tr['rpn_deltas'].contiguous()


def wall(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def rpn(debug=False):
    return ops.rpn_proposals(scores, deltas, P['anchors'], fh, fw, rp['anchor_stride'], ih, iw, rp['target_means'],
                             rp['target_stds'], tc['rpn']['nms_pre'], tc['rpn']['min_bbox_size'],
                             tc['rpn']['nms_iou_threshold'], tc['rpn']['max_per_img'], debug_topk=debug)


props, n_p, dbg = rpn(True)
torch.cuda.synchronize()
st = dbg[0, 8192 - 16: 8192 - 16 + 10].cpu().numpy().astype(np.int64)
print(f'rpn_proposals ({which}): n_props {int(n_p.item())}; whole pipeline (hist x2, compact, ranksort + decode, IoU matrix, '
      f'matrix NMS, proposals) {wall(rpn):.1f} us per call')
if st[9] > 0:     # rpn_matrix_nms_kernel finished the image: its own stamps
    print(f'  rpn_matrix_nms_kernel (one wavefront): resolution + outputs {(st[6] - st[5]) / 100.0:.1f} us')
else:
    for i, nm in enumerate(['load', 'select', 'compact', 'sort', 'decode', 'nms+out']):
        print(f'  {nm:10s} {(st[i + 1] - st[i]) / 100.0:8.1f} us')

pi = tr['per_image'][0]
rois, cls_raw, reg_raw = pi['rois'].contiguous(), pi['cls_raw'].contiguous(), pi['reg_raw'].contiguous()
bh = rh['bbox_head']


def det(debug=False):
    return ops.det_post(rois, cls_raw, reg_raw, n, ih, iw, bh['target_means'], bh['target_stds'], tc['rcnn']['score_thr'],
                        tc['rcnn']['nms_iou_threshold'], tc['rcnn']['max_per_img'], tr['n_props'][0:1], debug_scores=debug)


out = det(True)
torch.cuda.synchronize()
st = ops.det_post.last_stamps.cpu().numpy().astype(np.int64)
print(f'det_post: n_det {int(out[2].item())}, candidates over the score threshold {st[5]}; {wall(det):.1f} us per call')
for nm, a, c in (('decode', 0, 1), ('sort', 1, 2), ('offset', 2, 3), ('nms', 3, 6), ('out', 6, 4)):
    print(f'  {nm:10s} {(st[c] - st[a]) / 100.0:8.1f} us')
