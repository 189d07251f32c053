"""Oracle (oracle/fgn_ref_cpu.py) vs golden vectors produced by the reference's own
head files (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import torch
from torch import nn

from oracle import fgn_ref_cpu as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _rpn_weights(z):
    return {k.replace('__', '.'): torch.from_numpy(z[k]) for k in z.files if k.startswith('rpn_head')}


def test_ag_rpn_matches_reference(golden_dir):
    for name in ('ag_rpn.npz', 'ag_rpn_n1.npz'):
        z = _load(golden_dir, name)
        cls, reg = O.ag_rpn_forward(torch.from_numpy(z['qry']), torch.from_numpy(z['spp']),
                                    _rpn_weights(z), int(z['n_ways']), int(z['k_shots']))
        # identical op sequence on identical inputs -> bitwise
        assert np.array_equal(cls.numpy(), z['cls'])
        assert np.array_equal(reg.numpy(), z['reg'])


def test_count_modified_cls_bbox_matches_reference(golden_dir):
    z = _load(golden_dir, 'cls_bbox.npz')
    cls_raw, reg_raw = torch.from_numpy(z['cls_raw']), torch.from_numpy(z['reg_raw'])
    c3, r3 = O.count_modified_cls_bbox(7, cls_raw, reg_raw, 3)
    assert np.array_equal(c3.numpy(), z['cls_n3']) and np.array_equal(r3.numpy(), z['reg_n3'])
    c1, r1 = O.count_modified_cls_bbox(7, cls_raw[:7], reg_raw[:7], 1)
    assert np.array_equal(c1.numpy(), z['cls_n1']) and np.array_equal(r1.numpy(), z['reg_n1'])


def test_count_spp_matches_reference(golden_dir):
    z = _load(golden_dir, 'count_spp.npz')
    cfg = {'roi_head': {'featmap_stride': 16, 'roi_out_size': 7,
                        'shared_head': {'num_blocks': 0}}, 'backbone': {'bn_eps': 1e-5}}
    cat_mean, cat_mean_mp, _, _ = O.count_spp(
        torch.from_numpy(z['spp_fmaps']), z['spp_bboxes_xyxy'],
        torch.from_numpy(z['spp_isegmaps']), {}, cfg, int(z['n_ways']), int(z['k_shots']))
    assert np.array_equal(cat_mean.numpy(), z['cat_mean'])
    assert np.array_equal(cat_mean_mp.numpy(), z['cat_mean_mp'])


def test_relation_matches_reference(golden_dir):
    z = _load(golden_dir, 'relation.npz')
    # replay the torch.nn constructors the reference's init_cls_reg_shared_conv runs
    torch.manual_seed(int(z['seed_weights']))
    conv = nn.Conv2d(2048, 1024, kernel_size=(1, 1))
    nn.GroupNorm(32, 1024)
    nn.init.kaiming_normal_(conv.weight, nonlinearity='relu')
    assert abs(float(conv.weight.double().sum()) - float(z['conv_weight_sum'])) < 1e-6
    assert np.array_equal(conv.bias.detach().numpy(), z['conv_bias'])
    gi = torch.Generator().manual_seed(int(z['seed_inputs']))
    bbox_feats = torch.randn(3, 1024, 7, 7, generator=gi).abs()
    cat_mean = torch.randn(2, 3, 1024, 7, 7, generator=gi).abs()
    sd = {'roi_head.cls_reg_shared_conv.weight': conv.weight.detach(),
          'roi_head.cls_reg_shared_conv.bias': conv.bias.detach(),
          'roi_head.cls_reg_shared_conv_norm.weight': torch.from_numpy(z['gn_weight']),
          'roi_head.cls_reg_shared_conv_norm.bias': torch.from_numpy(z['gn_bias'])}
    cfg = {'roi_head': {'relation': {'gn_groups': 32, 'gn_eps': 1e-5}}}
    with torch.no_grad():
        rel = O.relation(bbox_feats, z['rois'], cat_mean, sd, cfg, 3)
    assert tuple(rel.shape) == tuple(z['out_shape'])
    assert np.array_equal(rel.reshape(-1)[::97].numpy(), z['out_sample'])
    assert abs(float(rel.double().sum()) - float(z['out_sum'])) < 1e-3


def test_mask_vector_gather_matches_reference(golden_dir):
    z = _load(golden_dir, 'mask_gather.npz')
    mp = torch.from_numpy(z['cat_mean_mp'])
    labels = [z['det_labels_0'], z['det_labels_1']]
    n_ways = mp.shape[1]
    gather = np.concatenate([labels[i] + n_ways * i for i in range(2)])
    vec = mp.view(2 * n_ways, -1, 1, 1)[torch.from_numpy(gather)]
    assert np.array_equal(vec.numpy(), z['spp_vecs_mask'])
    assert np.array_equal((torch.from_numpy(z['feats']) * vec).numpy(), z['mask_pred'])
    assert (z['labels_mask_0'] == 0).all()


def test_published_known_answers_of_the_upstream_box_code():
    """The parts of the path that live in mmdet 2.18 / mmcv 1.3.16 (absent from the image) are restated from their
    published algorithm; their docstrings carry known-answer examples, reproduced here:
    - mmdet/core/bbox/coder/delta_xywh_bbox_coder.py, `delta2bbox` example (means 0, stds 1, max_shape (32,32,3));
    - mmdet/core/anchor/anchor_generator.py, `AnchorGenerator([16],[1.],[1.],[9]).grid_anchors([(2,2)])`;
    - mmcv/ops/nms.py, `nms` example (7 boxes, IoU 0.6 -> 3 kept)."""
    from oracle import fgn_ref_cpu as O
    rois = np.array([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]], np.float32)
    deltas = np.array([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.], [0.7, -1.9, -0.5, 0.3]], np.float32)
    out = O.delta2bbox(rois, deltas, (0., 0., 0., 0.), (1., 1., 1., 1.), (32, 32, 3))
    want = np.array([[0.0000, 0.0000, 1.0000, 1.0000], [0.1409, 0.1409, 2.8591, 2.8591],
                     [0.0000, 0.3161, 4.1945, 0.6839], [5.0000, 5.0000, 5.0000, 5.0000]], np.float32)
    assert np.allclose(out, want, atol=5e-5)
    base = O.base_anchors((1.,), (1.,), 9)
    grid = O.grid_anchors(base, 2, 2, 16)
    assert np.array_equal(grid, np.array([[-4.5, -4.5, 4.5, 4.5], [11.5, -4.5, 20.5, 4.5],
                                          [-4.5, 11.5, 4.5, 20.5], [11.5, 11.5, 20.5, 20.5]], np.float32))
    boxes = np.array([[49.1, 32.4, 51.0, 35.9], [49.3, 32.9, 51.0, 35.3], [49.2, 31.8, 51.0, 35.4],
                      [35.1, 11.5, 39.1, 15.7], [35.6, 11.8, 39.3, 14.2], [35.3, 11.5, 39.9, 14.5],
                      [35.2, 11.7, 39.7, 15.7]], np.float32)
    scores = np.array([0.9, 0.9, 0.5, 0.5, 0.5, 0.4, 0.3], np.float32)
    dets, inds = O.nms(boxes, scores, 0.6)
    assert len(dets) == len(inds) == 3 and inds.tolist() == [0, 3, 4]
