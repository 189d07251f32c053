"""The training oracle against the golden vectors generated from the reference's own files
(tests/golden/make_golden_train.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import fgn_train_cpu as T

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_max_iou_assigner_matches_the_vendored_file():
    z = np.load(os.path.join(G, 'train_assign.npz'))
    for c in range(int(z['n_cases'])):
        ov = torch.from_numpy(z[f'c{c}_overlaps'])
        pos, neg, mn = (float(v) for v in z[f'c{c}_thr'])
        gi, mo = T.max_iou_assign(ov, pos, neg, mn, True)
        assert np.array_equal(gi.numpy(), z[f'c{c}_gt_inds']), c
        assert np.array_equal(mo.numpy(), z[f'c{c}_max_overlaps']), c
        # labels of the positives (AssignResult.labels)
        labels = np.full(len(gi), -1, np.int64)
        p = gi.numpy() > 0
        labels[p] = z[f'c{c}_gt_labels'][gi.numpy()[p] - 1]
        assert np.array_equal(labels, z[f'c{c}_labels']), c


def test_random_sampler_matches_the_vendored_file_under_the_same_seed():
    z = np.load(os.path.join(G, 'train_sample.npz'))
    for c in range(int(z['n_cases'])):
        gi = torch.from_numpy(z[f'c{c}_gt_inds'])
        num, frac = int(z[f'c{c}_cfg'][0]), float(z[f'c{c}_cfg'][1])
        torch.manual_seed(int(z[f'c{c}_seed']))
        s = T.random_sample(gi, torch.zeros(len(gi), 4), torch.zeros(3, 4), None, num, frac, False)
        # the vendored sampler returns the draw order; BaseSampler.sample sorts it with unique()
        assert np.array_equal(s['pos_inds'].numpy(), np.unique(z[f'c{c}_pos'])), c
        assert np.array_equal(s['neg_inds'].numpy(), np.unique(z[f'c{c}_neg'])), c


def test_bbox_head_targets_and_loss_match_the_reference():
    z = np.load(os.path.join(G, 'train_bbox_loss.npz'))
    t = lambda k: torch.from_numpy(z[k])
    s = dict(pos_bboxes=t('pos_bboxes'), neg_bboxes=t('neg_bboxes'), pos_gt_bboxes=t('pos_gt_bboxes'),
             pos_gt_labels=t('pos_gt_labels'))
    tg = T.bbox_targets_single(s, 3, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), -1)
    for a, k in zip(tg, ('labels', 'label_weights', 'bbox_targets', 'bbox_weights')):
        assert np.array_equal(a.numpy(), z[k]), k
    L = T.bbox_loss(t('cls_score'), t('bbox_pred'), *tg, 3)
    assert float(L['loss_cls']) == pytest.approx(float(z['loss_cls']), rel=1e-6)
    assert float(L['loss_bbox']) == pytest.approx(float(z['loss_bbox']), rel=1e-6)
    assert float(L['ACC-Unbalanced']) == pytest.approx(float(z['acc']), abs=1e-7)
    assert float(L['ACC-Balanced']) == pytest.approx(float(z['acc_balanced']), abs=1e-7)
    n_pos = len(z['pos_bboxes'])
    s0 = dict(pos_bboxes=s['pos_bboxes'][:0], neg_bboxes=s['neg_bboxes'], pos_gt_bboxes=s['pos_gt_bboxes'][:0],
              pos_gt_labels=s['pos_gt_labels'][:0])
    tg0 = T.bbox_targets_single(s0, 3, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2), -1)
    assert np.array_equal(tg0[0].numpy(), z['labels_nopos'])
    L0 = T.bbox_loss(t('cls_score')[n_pos:], t('bbox_pred')[n_pos:], *tg0, 3)
    assert float(L0['loss_cls']) == pytest.approx(float(z['loss_cls_nopos']), rel=1e-6)
    assert float(L0['loss_bbox']) == float(z['loss_bbox_nopos']) == 0.0


def test_rpn_gt_grouping_and_balancer_match_the_reference():
    z = np.load(os.path.join(G, 'train_rpn_grouping.npz'))
    n = int(z['n_ways'])
    gts = T.group_gt_by_class([torch.from_numpy(z['qry_bboxes_0']), torch.from_numpy(z['qry_bboxes_1'])],
                              [torch.from_numpy(z['qry_cat_ids_0']), torch.from_numpy(z['qry_cat_ids_1'])], n)
    assert len(gts) == int(z['n_groups'])
    for i, g in enumerate(gts):
        assert np.array_equal(g.numpy(), z[f'gt_{i}']), i
    assert list(z['meta_tags']) == [0, 0, 0, 1, 1, 1]          # image-major, one meta per guided pass
    # the 1/N balancer (fgn_ag_rpn_head.py:77-78)
    assert float(z['loss_rpn_cls']) == pytest.approx(float(z['loss_in'][0]) / n, rel=1e-6)
    assert float(z['loss_rpn_bbox']) == pytest.approx(float(z['loss_in'][1]) / n, rel=1e-6)


def test_mask_vector_gather_matches_the_reference():
    z = np.load(os.path.join(G, 'train_mask_gather.npz'))
    v = T.mask_vector_gather(torch.from_numpy(z['cat_mean_mp']),
                             [torch.from_numpy(z['pos_gt_labels_0']), torch.from_numpy(z['pos_gt_labels_1'])], 3)
    assert np.array_equal(v.numpy(), z['spp_vecs_mask'])
    assert list(z['loss_keys']) == ['loss_cls', 'loss_mask']


def test_published_assigner_docstring_example():
    """MaxIoUAssigner docstring (my_max_iou_assigner.py:86-91): expected_gt_inds == [1, 0]."""
    b = torch.Tensor([[0, 0, 10, 10], [10, 10, 20, 20]])
    g = torch.Tensor([[0, 0, 10, 9]])
    gi, _ = T.max_iou_assign(T.bbox_overlaps(g, b), 0.5, 0.5, 0.0, True)
    assert gi.tolist() == [1, 0]


def test_rpn_loss_matches_the_vendored_anchor_head():
    """AnchorHead.loss of my_anchor_head.py (targets, unmapping, num_total_samples, flattening) under seed 77."""
    from fgn_amd.config import fgn_r50_c4_config
    z = np.load(os.path.join(G, 'train_anchor_loss.npz'))
    cfg = fgn_r50_c4_config(3, 2)
    n = int(z['n_groups'])
    gts = [torch.from_numpy(z[f'gt_{i}']) for i in range(n)]
    torch.manual_seed(int(z['seed']))
    lc, lb = T.rpn_loss(torch.from_numpy(z['cls']), torch.from_numpy(z['reg']), gts,
                        [torch.from_numpy(z['img_hw'])] * n, cfg, torch.randperm)
    assert float(lc) == pytest.approx(float(z['loss_cls']), rel=1e-6)
    assert float(lb) == pytest.approx(float(z['loss_bbox']), rel=1e-6)


def test_default_train_and_test_constants_equal_the_reference_configs():
    """config.fgn_r50_c4_config's train_cfg / test_cfg against the dicts of the reference's own model configs (dumped
    as data by make_golden_train.py), through the same normalisation `FGN(..., train_cfg=, test_cfg=)` applies."""
    import json
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import normalise_config
    z = json.load(open(os.path.join(G, 'train_cfg_reference.json')))
    ours = fgn_r50_c4_config(3, 3)
    got = normalise_config(3, 3, test_cfg=z['test_cfg'], train_cfg=z['train_cfg'])
    assert got['train_cfg'] == ours['train_cfg']
    assert got['test_cfg'] == ours['test_cfg']
    # and the values themselves, spelled out
    assert got['train_cfg']['rpn']['num'] == 64 and got['train_cfg']['rcnn']['num'] == 128
    assert got['train_cfg']['rpn_proposal'] == dict(nms_pre=12000, max_per_img=2000, nms_iou_threshold=0.7, min_bbox_size=0)
