"""Helpers shared by the CPU and GPU tests of tests/golden/fgn_glue.npz (the reference's own fgn.py glue)."""
import os

import numpy as np


def glue_inputs():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_fgn', os.path.join(
        os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_golden_fgn.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)            # pure helpers at import time: nothing reads /root/reference
    return m.make_inputs()


def glue_expected(z, prefix='server__'):
    """The result dicts stored by make_golden_fgn.py as a list of {key: ndarray | list of RLE dicts}."""
    out = []
    for i in range(int(z[prefix + 'n_out'])):
        one = {}
        for k in z[prefix + 'out_keys']:
            k = str(k)
            if k.endswith('_rle'):
                one[k] = [{'size': z[f'{prefix}out{i}__{k}__{j}__size'].tolist(),
                           'counts': z[f'{prefix}out{i}__{k}__{j}__counts'].tobytes()}
                          for j in range(int(z[f'{prefix}out{i}__{k}__n']))]
            else:
                one[k] = z[f'{prefix}out{i}__{k}']
        out.append(one)
    return out


def assert_results_equal(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert list(g) == list(w), (list(g), list(w))            # same keys in the same order
        for k in w:
            if k.endswith('_rle'):
                assert g[k] == w[k], k
            else:
                assert isinstance(g[k], np.ndarray) and g[k].dtype == w[k].dtype and g[k].shape == w[k].shape, \
                    (k, g[k].dtype, w[k].dtype, g[k].shape, w[k].shape)
                assert np.array_equal(g[k], w[k]), k


def reference_model_cfg():
    """``cfg.model`` of the reference minus its ``type`` key: the mmcv dicts of fgn_r50_c4_densecl.py:13-186 as
    ``build_detector`` passes them to the detector class, plus the ``n_ways`` / ``k_shots`` main.py injects."""
    return dict(
        n_ways=3, k_shots=3,
        backbone=dict(type='ResNet', depth=50, num_stages=4, strides=(1, 2, 2, 2), out_indices=(2,),
                      frozen_stages=4, norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True,
                      style='pytorch'),
        rpn_head=dict(type='AGRPNHead', num_convs=1, in_channels=1024, feat_channels=1024,
                      anchor_generator=dict(type='AnchorGenerator', scales=[2, 4, 8, 16, 32],
                                            ratios=[0.5, 1.0, 2.0], strides=[16]),
                      bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0],
                                      target_stds=[1.0, 1.0, 1.0, 1.0])),
        roi_head=dict(type='FGNRoIHead', shared_head=None,
                      bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                              roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                              out_channels=1024, featmap_strides=[16]),
                      bbox_head=dict(type='FGNBBoxHead', with_avg_pool=True, roi_feat_size=7, in_channels=1024,
                                     num_classes=1, reg_class_agnostic=False,
                                     bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.],
                                                     target_stds=[0.1, 0.1, 0.2, 0.2])),
                      mask_head=dict(type='FCNMaskHead', num_convs=4, in_channels=1024, conv_out_channels=256,
                                     num_classes=1, class_agnostic=True)),
        test_cfg=dict(rpn=dict(nms_pre=6000, nms=dict(type='nms', iou_threshold=0.7), max_per_img=300,
                               min_bbox_size=0),
                      rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100,
                                mask_thr_binary=0.5)))
