"""Model hyper-parameters of the FGN inference path.

Plain-dict restatement of the constants the reference keeps in its mmcv config
(reference: subprojects/sp02_omniiseg_fgn_mmdet/fgn_r50_c4_densecl.py:13-186).
``train_cfg`` flattens the assigner / sampler dicts of fgn_r50_c4_densecl.py:131-171 (MaxIoUAssigner +
RandomSampler for both stages); the loss types are the config's (sigmoid CE + SmoothL1(beta 1) for the
AG-RPN, softmax CE + SmoothL1 for the box head, per-pixel BCE for the class-agnostic mask head).
"""
from __future__ import annotations

import copy


def fgn_r50_c4_config(n_ways: int = 3, k_shots: int = 3) -> dict:
    """Config of the DenseCL ResNet-50-C4 FGN (fgn_r50_c4_densecl.py:13-186)."""
    return dict(
        type='FGN',
        n_ways=n_ways,
        k_shots=k_shots,
        backbone=dict(
            type='ResNet', depth=50, block='bottleneck',
            # layer4 is deleted at run time (main.py:403-405); out_indices=(2,)
            stage_blocks=(3, 4, 6), stage_planes=(64, 128, 256),
            strides=(1, 2, 2), stem_channels=64, style='pytorch',
            deep_stem=False, avg_down=False, norm='BN', gn_groups=32,
            norm_eval=True, bn_eps=1e-5),
        rpn_head=dict(
            type='AGRPNHead', in_channels=1024, feat_channels=1024,
            anchor_scales=(2, 4, 8, 16, 32), anchor_ratios=(0.5, 1.0, 2.0),
            anchor_stride=16,
            target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.)),
        roi_head=dict(
            type='FGNRoIHead',
            roi_out_size=7, roi_sampling_ratio=0, featmap_stride=16,
            shared_head=dict(inplanes=1024, planes=512, num_blocks=3),
            relation=dict(in_channels=2048, out_channels=1024, gn_groups=32,
                          gn_eps=1e-5),
            bbox_head=dict(in_channels=1024, num_classes=1,
                           target_means=(0., 0., 0., 0.),
                           target_stds=(0.1, 0.1, 0.2, 0.2)),
            mask_head=dict(num_convs=4, in_channels=1024,
                           conv_out_channels=256, num_classes=1)),
        test_cfg=dict(
            rpn=dict(nms_pre=6000, nms_iou_threshold=0.7, max_per_img=300,
                     min_bbox_size=0),
            rcnn=dict(score_thr=0.05, nms_iou_threshold=0.5, max_per_img=100,
                      mask_thr_binary=0.5)),
        train_cfg=dict(
            rpn=dict(pos_iou_thr=0.5, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True,
                     num=64, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False,
                     allowed_border=0, pos_weight=-1),
            rpn_proposal=dict(nms_pre=12000, max_per_img=2000, nms_iou_threshold=0.7, min_bbox_size=0),
            rcnn=dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=True,
                      num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True,
                      mask_size=14, pos_weight=-1)),
    )


def fgn_r18_c4_config(n_ways: int = 3, k_shots: int = 1) -> dict:
    """cfg2 of BASELINE.json words the OMNIISEG case with a ResNet-18 backbone.  The reference has no such config
    ("This file is for 50 layers ONLY", fgn_r50_c4_densecl.py:17-18; 1024 channels are hard-coded in its heads,
    fgn_roi_head.py:210-213, 241-243), so this is a build extension without reference semantics:
    mmdet's ResNet-18 (BasicBlock, stage_blocks (2, 2, 2), C4 = layer3 = 256 channels) with every head width taken
    from the C4 width the way the R50 config takes it from 1024: RPN 256 -> 256, shared_head 256 -> 128 -> 256,
    relation 512 -> 256 with GroupNorm(32, 256) (8 channels per group), box / mask heads on 256 channels."""
    cfg = fgn_r50_c4_config(n_ways, k_shots)
    cfg['backbone'].update(depth=18, block='basic', stage_blocks=(2, 2, 2), stage_planes=(64, 128, 256))
    c = 256
    cfg['rpn_head'].update(in_channels=c, feat_channels=c)
    cfg['roi_head']['shared_head'].update(inplanes=c, planes=c // 2)
    cfg['roi_head']['relation'].update(in_channels=2 * c, out_channels=c, gn_groups=32)
    cfg['roi_head']['bbox_head'].update(in_channels=c)
    cfg['roi_head']['mask_head'].update(in_channels=c, conv_out_channels=256)
    return cfg


def fgn_r50_c4_scratch_config(n_ways: int = 3, k_shots: int = 3) -> dict:
    """The from-scratch backbone variant (fgn_r50_c4_scratch.py:5-30): 3-conv deep stem, average-pool
    shortcuts, GroupNorm(32) instead of frozen BatchNorm.  Heads and test_cfg are the same."""
    cfg = fgn_r50_c4_config(n_ways, k_shots)
    # ref_num_stages: fgn_r50_c4_scratch.py:12 builds THREE stages (no layer4 module at all); the DenseCL config builds four
    # and main.py:402-405 only stops walking the last one (fgn_amd.train.reference_param_order)
    cfg['backbone'].update(deep_stem=True, avg_down=True, norm='GN', gn_groups=32, ref_num_stages=3)
    return cfg


def tiny_config(n_ways: int = 3, k_shots: int = 1, width_div: int = 8, scratch: bool = False) -> dict:
    """A narrow variant (all channel widths divided) for fast CPU tests.

    Not a reference configuration: same topology, smaller widths, so the
    oracle and the host logic can be exercised in seconds on CPU.
    """
    cfg = (fgn_r50_c4_scratch_config if scratch else fgn_r50_c4_config)(n_ways, k_shots)
    d = width_div
    cfg['backbone'].update(stage_planes=tuple(p // d for p in (64, 128, 256)),
                           stem_channels=64 // d)
    if scratch:     # GroupNorm(32) needs >= 32 channels everywhere: keep the stem at full width
        cfg['backbone'].update(stem_channels=64, gn_groups=min(32, 64 // d))
    c = 1024 // d
    cfg['rpn_head'].update(in_channels=c, feat_channels=c)
    cfg['roi_head']['shared_head'].update(inplanes=c, planes=c // 2)
    cfg['roi_head']['relation'].update(in_channels=2 * c, out_channels=c,
                                       gn_groups=max(1, 32 // d))
    cfg['roi_head']['bbox_head'].update(in_channels=c)
    cfg['roi_head']['mask_head'].update(in_channels=c,
                                        conv_out_channels=256 // d)
    return cfg


def with_caps(cfg: dict, nms_pre=None, rpn_max=None, det_max=None) -> dict:
    cfg = copy.deepcopy(cfg)
    if nms_pre is not None:
        cfg['test_cfg']['rpn']['nms_pre'] = nms_pre
    if rpn_max is not None:
        cfg['test_cfg']['rpn']['max_per_img'] = rpn_max
    if det_max is not None:
        cfg['test_cfg']['rcnn']['max_per_img'] = det_max
    return cfg
