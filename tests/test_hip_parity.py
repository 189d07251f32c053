"""Stage parity at full width: the reference's own golden outputs through the HIP kernels (relation, mask
gather), and the HIP shared_head / mask-head stacks against the oracle on >= 100 RoIs of 1024 channels."""
import os

import numpy as np
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _rel_err(got, ref):
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)


def test_relation_kernel_matches_reference_golden(golden_dir):
    """tests/golden/relation.npz = the reference's own ``count_one_roi_by_n_spp`` (fgn_roi_head.py:253-279) at
    C = 1024 with the conv its own ``init_cls_reg_shared_conv`` built (seed replayed as in test_oracle_golden).
    HIP: split relation conv (Wq on the RoIs, Ws + bias on the class means) + fused GN/ReLU kernel."""
    from fgn_amd import ops
    z = np.load(os.path.join(golden_dir, 'relation.npz'))
    torch.manual_seed(int(z['seed_weights']))
    conv = nn.Conv2d(2048, 1024, kernel_size=(1, 1))
    nn.GroupNorm(32, 1024)
    nn.init.kaiming_normal_(conv.weight, nonlinearity='relu')
    assert abs(float(conv.weight.double().sum()) - float(z['conv_weight_sum'])) < 1e-6
    gi = torch.Generator().manual_seed(int(z['seed_inputs']))
    bbox_feats = torch.randn(3, 1024, 7, 7, generator=gi).abs()
    cat_mean = torch.randn(2, 3, 1024, 7, 7, generator=gi).abs()
    w = conv.weight.detach()
    wq = ops.pack_conv(w[:, :1024].contiguous()).to('cuda')
    ws = ops.pack_conv(w[:, 1024:].contiguous(), bias=conv.bias.detach()).to('cuda')
    Q = ops.conv2d(_nhwc(bbox_feats).cuda(), wq)
    S = ops.conv2d(_nhwc(cat_mean.view(6, 1024, 7, 7)).cuda(), ws)
    rois = torch.from_numpy(z['rois']).cuda()
    rel = torch.zeros(9, 7, 7, 1024, device='cuda')
    fc_w = torch.zeros(6, 1024, device='cuda')
    fc_w[0] = 1.0                                            # cls column 0 = sum_c mean_hw(rel): a checksum
    cls, _ = ops.relation_gn_head(Q, S, rois, torch.from_numpy(z['gn_weight']).cuda(),
                                  torch.from_numpy(z['gn_bias']).cuda(), fc_w, torch.zeros(6, device='cuda'), 3, 32,
                                  1e-5, rel_out=rel)
    got = _nchw(rel.cpu())
    assert tuple(got.shape) == tuple(z['out_shape'])
    sample = got.reshape(-1)[::97].numpy()
    err = np.abs(sample - z['out_sample']).max() / np.abs(z['out_sample']).max()
    print(f'[parity relation golden C=1024] rel err of the sampled outputs {err:.2e}')
    assert err <= 2e-5
    assert abs(float(got.double().sum()) - float(z['out_sum'])) <= 2e-6 * float(z['out_sum'])
    # the fused pooled+FC output agrees with the map it did not materialise
    assert abs(float(cls[:, 0].double().sum()) * 49 - float(z['out_sum'])) <= 1e-5 * float(z['out_sum'])


def test_mask_gather_matches_reference_golden(golden_dir):
    """tests/golden/mask_gather.npz = the reference's label -> support-vector gather (fgn_roi_head.py:707-714)
    and guidance multiply of ``_mask_forward`` (379).  HIP: gather kernel (bitwise) and the multiply as it is
    fused into mask conv 0 - register-staged direct kernel (identity 1x1: bitwise) and Winograd input
    transform (identity 3x3 centre tap: 1e-6)."""
    from fgn_amd import ops
    z = np.load(os.path.join(golden_dir, 'mask_gather.npz'))
    mp = torch.from_numpy(z['cat_mean_mp'])                      # [2,3,8,1,1]
    b, n, c = mp.shape[:3]
    labels = torch.from_numpy(np.concatenate([z['det_labels_0'], z['det_labels_1']])).long()
    img_idx = torch.tensor([0] * len(z['det_labels_0']) + [1] * len(z['det_labels_1']), dtype=torch.float32)
    rois = torch.zeros(len(labels), 5)
    rois[:, 0] = img_idx
    vec = ops.gather_support_vectors(mp.reshape(b * n, c).contiguous().cuda(), labels.cuda(), rois.cuda(), n)
    assert np.array_equal(vec.cpu().numpy(), z['spp_vecs_mask'].reshape(len(labels), c))
    # the multiply, fused into the first mask conv: pad 8 -> 32 channels (zeros), identity weights
    feats = torch.from_numpy(z['feats'])                          # [6,8,7,7]
    d = feats.shape[0]
    f32 = torch.zeros(d, 32, 7, 7)
    f32[:, :c] = feats
    v32 = torch.zeros(d, 32)
    v32[:, :c] = vec.cpu()
    want = torch.from_numpy(z['mask_pred'])                       # feats * vec, [6,8,7,7]
    eye1 = ops.pack_conv(torch.eye(32).reshape(32, 32, 1, 1)).to('cuda')
    y = ops.conv2d(_nhwc(f32).cuda(), eye1, in_scale=v32.cuda())
    assert torch.equal(_nchw(y.cpu())[:, :c], want)
    w3 = torch.zeros(32, 32, 3, 3)
    w3[:, :, 1, 1] = torch.eye(32)
    y3 = ops.conv3x3_winograd(_nhwc(f32).cuda(), ops.pack_winograd(w3).to('cuda'), in_scale=v32.cuda())
    assert _rel_err(_nchw(y3.cpu())[:, :c], want) <= 1e-6


def _full_model():
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    cfg = fgn_r50_c4_config(3, 3)
    sd = init_state_dict(cfg, 0)
    return cfg, sd, FGN(3, 3, state_dict=sd)


def _rois(g, r, h, w):
    x0 = torch.rand(r, generator=g) * (w * 0.8)
    y0 = torch.rand(r, generator=g) * (h * 0.8)
    bw = torch.rand(r, generator=g) * (w * 0.5) + 4
    bh = torch.rand(r, generator=g) * (h * 0.5) + 4
    return torch.stack([torch.zeros(r), x0, y0, (x0 + bw).clamp(max=w), (y0 + bh).clamp(max=h)], 1)


@pytest.mark.parametrize('commute', [True, False])
def test_shared_head_stage_matches_oracle_full_width(commute):
    """RoIAlign + ``shared_head`` (fgn_roi_head.py:331-336; 3 bottlenecks 1024 -> 512 -> 1024 at 7x7) on 128 RoIs
    of a 50x84x1024 map: the product's ``_roi_feats`` (Winograd 3x3; with ``commute`` the first 1x1 conv is
    taken on the feature map before the pooling) against ``O.roi_align`` + ``O.shared_head``."""
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    cfg, sd, model = _full_model()
    model.use_roi_commute = commute
    model._pack(torch.device('cuda', 0))
    g = torch.Generator().manual_seed(31)
    fmap = torch.randn(1, 1024, 50, 84, generator=g).relu()         # a C4 map is a ReLU output
    rois = _rois(g, 128, 800, 1333)
    rois[0, 1:] = torch.tensor([0., 0., 1333., 800.])
    rois[1, 1:] = torch.tensor([640., 400., 640.5, 400.25])         # sub-pixel RoI
    ref_pre = O.roi_align(fmap, rois.numpy(), 7, 1 / 16, 0, True)
    ref = O.shared_head(ref_pre, sd, cfg)
    x = _nhwc(fmap).cuda()
    g_map = ops.conv2d(x, model._P['sh0_lin']) if commute else None
    assert (model._P['sh0_lin'] is not None) == commute
    cnt = torch.tensor([128], dtype=torch.int32, device='cuda')
    pre, got = model._roi_feats(x, g_map, rois.cuda(), cnt)
    e_pre, e = _rel_err(_nchw(pre.cpu()), ref_pre), _rel_err(_nchw(got.cpu()), ref)
    print(f'[parity shared_head 128 RoIs x 1024ch, commute={commute}] rel err RoIAlign {e_pre:.2e}, shared_head {e:.2e}')
    assert e_pre <= 2e-6 and e <= 2e-5


def test_mask_head_stage_matches_oracle_full_width():
    """``_mask_forward`` after the shared_head (fgn_roi_head.py:379-380): guidance multiply + FCNMaskHead on 100
    RoIs of 1024 channels; logits within 1e-4 of their range, probabilities within 1e-4 absolute (north_star)."""
    from oracle import fgn_ref_cpu as O
    cfg, sd, model = _full_model()
    model._pack(torch.device('cuda', 0))
    g = torch.Generator().manual_seed(32)
    mf = torch.randn(100, 1024, 7, 7, generator=g).relu()
    vec = torch.randn(100, 1024, generator=g).abs() * 0.5
    ref_logits = O.mask_head_forward(mf * vec[:, :, None, None], sd, cfg)[:, 0]
    ref_prob = torch.from_numpy(O.sigmoid32(ref_logits.numpy()))
    cnt = torch.tensor([100], dtype=torch.int32, device='cuda')
    logits, prob = model._mask_head(_nhwc(mf).cuda(), vec.cuda(), cnt)
    e_l = _rel_err(logits.cpu(), ref_logits)
    e_p = (prob.cpu() - ref_prob).abs().max().item()
    print(f'[parity mask head 100 RoIs x 1024ch] rel err logits {e_l:.2e} (|logit| <= {ref_logits.abs().max():.1f}), '
          f'max |d prob| {e_p:.2e}')
    assert e_l <= 2e-5 and e_p <= 1e-4


def test_input_shim_and_result_packing_match_reference_fgn_py(golden_dir):
    """tests/golden/fgn_glue.npz = the reference's own fgn.py (``modify_input``, ``get_img_metas``, the packing loop
    of ``simple_test``: fgn.py:79-123, 240-303) run around recorder heads.  HIP side: ``FGN._support_front`` must
    hand the heads the same XYXY support boxes / mask views without touching the caller's tensors, and
    ``FGN.pack_results`` must turn the same head outputs (detections, labels, dense masks -> device RLE) and
    passthrough inputs into the same result dicts: keys, key order, dtypes, shapes, values, RLE strings."""
    from fgn_amd import ops
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    from _glue import assert_results_equal, glue_expected, glue_inputs
    ins = glue_inputs()
    before = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in ins.items()}
    z = np.load(os.path.join(golden_dir, 'fgn_glue.npz'))
    cfg = tiny_config(3, 2, width_div=2)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    dev = torch.device('cuda', torch.cuda.current_device())
    model._pack(dev)
    main = torch.cuda.current_stream()
    B = ins['qry_img'].shape[0]
    # ---- input side (modify_input + the views of fgn.py:213-218)
    sc = model._support_front(ins['spp_imgs'], ins['spp_bboxes'], ins['spp_isegmaps'], B, dev, main)
    assert np.array_equal(sc['spp_xyxy'].cpu().numpy(), z['server__roi_spp_bboxes'].reshape(-1, 4))
    assert np.array_equal(sc['spp_masks'].cpu().numpy().astype(bool), z['server__roi_spp_isegmaps'][:, 0])
    assert tuple(sc['spp_fmaps'].shape[:1]) == (int(z['server__extract_shapes'][1][0]),)
    # ---- output side: the recorder heads' outputs as the device tensors detect_device produces
    D, H, W = cfg['test_cfg']['rcnn']['max_per_img'], 24, 40
    outs = []
    for i in range(B):
        det, lab, seg = z[f'server__head_det{i}'], z[f'server__head_lab{i}'], z[f'server__head_seg{i}']
        n = len(det)
        d = dict(det_bboxes=torch.zeros(D, 5, device=dev), det_labels=torch.zeros(D, dtype=torch.int64, device=dev),
                 n_dets=torch.tensor([n], dtype=torch.int32, device=dev), mask_prob=torch.zeros(D, 14, 14, device=dev),
                 rle_bytes=torch.zeros(D, ops.RLE_BYTE_CAP, dtype=torch.uint8, device=dev),
                 rle_len=torch.zeros(D, dtype=torch.int32, device=dev),
                 rle_overflow=torch.zeros(D, dtype=torch.int32, device=dev), img_hw=(H, W))
        if n:
            d['det_bboxes'][:n] = torch.from_numpy(det).to(dev)
            d['det_labels'][:n] = torch.from_numpy(lab).to(dev)
            rb, rl, ro = ops.dense_mask_rle(torch.from_numpy(seg).to(dev))
            d['rle_bytes'][:n], d['rle_len'][:n], d['rle_overflow'][:n] = rb, rl, ro
        outs.append(d)
    _, gt_rle, uploaded = model._upload({'qry_img': ins['qry_img']}, ins['qry_isegmaps'], dev, main)
    for d, g in zip(outs, gt_rle):
        d['gt_rle'] = g
    model._start_download(outs, main, uploaded)
    got = model.pack_results(outs, B, **{k: ins[k] for k in (
        'qry_bboxes', 'qry_cat_ids', 'qry_isegmaps', 'img_shape', 'qry_child_idx', 'cats_ids_to_sample_real',
        'spp_insts_ids', 'idx')})
    assert_results_equal(got, glue_expected(z))
    # SERVER semantics: nothing the caller owns was modified
    for k, v in ins.items():
        for a, b in zip(v if isinstance(v, list) else [v], before[k] if isinstance(v, list) else [before[k]]):
            assert torch.equal(a, b), k
