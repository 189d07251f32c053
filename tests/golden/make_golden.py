#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own head files.

Run in the build container only (needs /root/reference; the GPU box never sees
it):   python tests/golden/make_golden.py

The reference's FGN-specific algebra lives in
  subprojects/sp02_omniiseg_fgn_mmdet/fgn_ag_rpn_head.py   (AGRPNHead.forward_single)
  subprojects/sp02_omniiseg_fgn_mmdet/fgn_roi_head.py      (FGNRoIHead.count_spp,
      count_one_roi_by_n_spp, count_modified_cls_bbox, _mask_forward, simple_test)
Those files import mmdet / mmcv / torchvision, which are neither in the reference
tree nor installed here (ordinary ModuleNotFoundError).  The
third-party *packages* are replaced by empty module objects whose classes are
bare ``torch.nn.Module`` bases, so that the reference's own files import and
their own pure-torch methods run unmodified.  Nothing from the reference is
copied: this script imports it, feeds seeded inputs and stores outputs.

What the stubs provide is stated with each golden; anything computed by a stub
(the RPN conv layers, roi_align) is injected from torch.nn / the oracle and is
therefore NOT pinned by these vectors -- only the reference's own algebra is.

Fixtures are data only: seeds + small tensors + strided samples of big outputs.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)


def _stub_third_party():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    ident_deco = lambda *a, **k: (lambda f: f)

    class _Base(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mmdet = mod('mmdet')
    models = mod('mmdet.models')
    det = mod('mmdet.models.detectors')
    det.TwoStageDetector = type('TwoStageDetector', (_Base,), {})
    builder = mod('mmdet.models.builder')
    builder.DETECTORS = builder.HEADS = builder.MODELS = _Registry()
    dense = mod('mmdet.models.dense_heads')
    dense.RPNHead = type('RPNHead', (_Base,), {})
    roi_heads = mod('mmdet.models.roi_heads')
    roi_heads.BBoxHead = type('BBoxHead', (_Base,), {})
    roi_heads.StandardRoIHead = type('StandardRoIHead', (_Base,), {})
    backbones = mod('mmdet.models.backbones')
    resnet = mod('mmdet.models.backbones.resnet')
    resnet.Bottleneck = type('Bottleneck', (_Base,), {'expansion': 4})
    utils = mod('mmdet.models.utils')
    utils.ResLayer = type('ResLayer', (_Base,), {})
    core = mod('mmdet.core')
    for n in ('encode_mask_results', 'BitmapMasks', 'bbox2result', 'bbox2roi',
              'build_assigner', 'build_sampler'):
        setattr(core, n, None)
    mod('mmdet.core.bbox')
    samplers = mod('mmdet.core.bbox.samplers')
    samplers.RandomSampler = object
    mod('mmcv')
    runner = mod('mmcv.runner')
    runner.auto_fp16 = ident_deco
    runner.force_fp32 = ident_deco
    mod('torchvision')
    tv_ops = mod('torchvision.ops')
    tv_ops.roi_align = None          # injected per golden below
    return tv_ops


def main():
    tv_ops = _stub_third_party()
    sys.path.insert(0, REF)
    from subprojects.sp02_omniiseg_fgn_mmdet import fgn_ag_rpn_head as ref_rpn
    from subprojects.sp02_omniiseg_fgn_mmdet import fgn_roi_head as ref_roi
    from oracle import fgn_ref_cpu as O

    g = torch.Generator().manual_seed(20261003)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ---- G1: AGRPNHead.forward_single (fgn_ag_rpn_head.py:26-118) -------------
    # stub: base RPNHead.forward_single = the three conv layers, given weights.
    B, N, K, C, A = 2, 3, 2, 32, 15
    w = {'rpn_head.rpn_conv.weight': rn(C, C, 3, 3) * 0.05, 'rpn_head.rpn_conv.bias': rn(C) * 0.1,
         'rpn_head.rpn_cls.weight': rn(A, C, 1, 1) * 0.3, 'rpn_head.rpn_cls.bias': rn(A) * 0.1,
         'rpn_head.rpn_reg.weight': rn(4 * A, C, 1, 1) * 0.1, 'rpn_head.rpn_reg.bias': rn(4 * A) * 0.1}
    sys.modules['mmdet.models.dense_heads'].RPNHead.forward_single = \
        lambda self, x: O.rpn_layers(x, w)
    head = ref_rpn.AGRPNHead()
    head.n_ways, head.k_shots = N, K
    qry = rn(B, C, 6, 7).abs()
    spp = rn(B * N * K, C, 4, 4).abs()
    cls, reg = head.forward_single(qry, spp)
    np.savez_compressed(os.path.join(OUT, 'ag_rpn.npz'), qry=qry.numpy(), spp=spp.numpy(),
                        cls=cls.numpy(), reg=reg.numpy(), n_ways=N, k_shots=K,
                        **{k.replace('.', '__'): v.numpy() for k, v in w.items()})
    # N == 1 branch
    head.n_ways, head.k_shots = 1, 2
    spp1 = rn(B * 1 * 2, C, 4, 4).abs()
    cls1, reg1 = head.forward_single(qry, spp1)
    np.savez_compressed(os.path.join(OUT, 'ag_rpn_n1.npz'), qry=qry.numpy(), spp=spp1.numpy(),
                        cls=cls1.numpy(), reg=reg1.numpy(), n_ways=1, k_shots=2,
                        **{k.replace('.', '__'): v.numpy() for k, v in w.items()})

    # ---- bare FGNRoIHead (constructor bypassed: it builds mmdet modules) -------
    rh = ref_roi.FGNRoIHead.__new__(ref_roi.FGNRoIHead)
    nn.Module.__init__(rh)
    rh.n_ways, rh.k_shots = 3, 2

    # ---- G2: count_modified_cls_bbox (fgn_roi_head.py:302-326) ----------------
    cls_raw, reg_raw = rn(7 * 3, 2), rn(7 * 3, 4)
    c3, r3 = rh.count_modified_cls_bbox(7, cls_raw, reg_raw)
    rh.n_ways = 1
    c1, r1 = rh.count_modified_cls_bbox(7, cls_raw[:7], reg_raw[:7])
    rh.n_ways = 3
    np.savez_compressed(os.path.join(OUT, 'cls_bbox.npz'), cls_raw=cls_raw.numpy(),
                        reg_raw=reg_raw.numpy(), cls_n3=c3.numpy(), reg_n3=r3.numpy(),
                        cls_n1=c1.numpy(), reg_n1=r1.numpy())

    # ---- G3: count_spp (fgn_roi_head.py:419-449) ------------------------------
    # stub: torchvision.ops.roi_align := oracle roi_align(aligned=False,
    # sampling_ratio=-1); shared head disabled.  Pins the in-place /16 box scaling,
    # the list-of-boxes calling convention and the two reductions.
    def tv_roi_align(inp, boxes, output_size):
        rois = np.concatenate([np.concatenate([np.full((len(b), 1), i, np.float32),
                                               b.detach().numpy()], 1)
                               for i, b in enumerate(boxes)], 0)
        return O.roi_align(inp, rois, output_size, 1.0, -1, aligned=False)
    ref_roi.roi_align = tv_roi_align
    type(rh).with_shared_head = property(lambda self: False)
    Bs, S, Cs = 2, 64, 16
    nk = rh.n_ways * rh.k_shots
    spp_fmaps = rn(Bs * nk, Cs, S // 16, S // 16).abs()
    boxes = torch.tensor([[8., 10., 50., 56.]]).repeat(Bs * nk, 1) + rn(Bs * nk, 4) * 3
    spp_boxes = boxes.view(-1, 1, 4).clone()
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing='ij')
    masks = torch.stack([((yy - 32 - i) ** 2 + (xx - 30) ** 2) < (18 + i) ** 2
                         for i in range(Bs * nk)]).view(-1, 1, S, S)
    rh.count_spp(spp_fmaps, spp_boxes.clone(), masks)
    np.savez_compressed(os.path.join(OUT, 'count_spp.npz'), spp_fmaps=spp_fmaps.numpy(),
                        spp_bboxes_xyxy=boxes.numpy(), spp_isegmaps=masks.numpy(),
                        cat_mean=rh.spp_fmaps_roi_aligned_cat_mean.numpy(),
                        cat_mean_mp=rh.spp_fvecs_roi_aligned_cat_mean_mp.numpy(),
                        n_ways=3, k_shots=2)

    # ---- G4: count_one_roi_by_n_spp (fgn_roi_head.py:240-279) -----------------
    # conv + GroupNorm are built by the reference's own init_cls_reg_shared_conv
    # under a fixed global seed; the test replays the same torch.nn constructors
    # under the same seed and checks a checksum before use.
    torch.manual_seed(4242)
    rh.init_cls_reg_shared_conv()
    with torch.no_grad():
        rh.cls_reg_shared_conv_norm.weight.copy_(0.75 + 0.5 * torch.rand(1024, generator=g))
        rh.cls_reg_shared_conv_norm.bias.copy_(0.1 * rn(1024))
    R = 3
    gi = torch.Generator().manual_seed(777)
    bbox_feats = torch.randn(R, 1024, 7, 7, generator=gi).abs()
    cat_mean = torch.randn(2, 3, 1024, 7, 7, generator=gi).abs()
    rois = torch.tensor([[0, 1, 2, 30, 40], [1, 5, 5, 60, 20], [1, 0, 0, 10, 10]], dtype=torch.float32)
    rh.spp_fmaps_roi_aligned_cat_mean = cat_mean
    with torch.no_grad():
        amount, rel = rh.count_one_roi_by_n_spp(bbox_feats, rois)
    assert amount == R
    np.savez_compressed(
        os.path.join(OUT, 'relation.npz'), seed_weights=4242, seed_inputs=777, rois=rois.numpy(),
        gn_weight=rh.cls_reg_shared_conv_norm.weight.detach().numpy(),
        gn_bias=rh.cls_reg_shared_conv_norm.bias.detach().numpy(),
        conv_weight_sum=float(rh.cls_reg_shared_conv.weight.double().sum()),
        conv_weight_abs=float(rh.cls_reg_shared_conv.weight.double().abs().sum()),
        conv_bias=rh.cls_reg_shared_conv.bias.detach().numpy(),
        out_sample=rel.reshape(-1)[::97].numpy(), out_sum=float(rel.double().sum()),
        out_shape=np.array(rel.shape))

    # ---- G5: label -> support-vector gather (fgn_roi_head.py:704-718) and the
    # guidance multiply of _mask_forward (fgn_roi_head.py:360-382) --------------
    # stubs: count_spp / simple_test_bboxes return fixed tensors; the RoI extractor,
    # shared head and mask head are identities, so mask_pred == mask_feats * vec.
    rh.n_ways, rh.k_shots = 3, 2
    mp = rn(2, 3, 8, 1, 1)
    det_labels = [torch.tensor([2, 0, 1, 1]), torch.tensor([0, 2])]
    det_bboxes = [rn(4, 5), rn(2, 5)]
    feats = rn(6, 8, 7, 7)
    captured = {}
    type(rh).with_bbox = property(lambda self: True)
    type(rh).with_mask = property(lambda self: True)
    rh.count_spp = lambda *a, **k: setattr(rh, 'spp_fvecs_roi_aligned_cat_mean_mp', mp)
    rh.simple_test_bboxes = lambda *a, **k: (det_bboxes, det_labels)

    def fake_simple_test_mask(x, metas, db, dl, rescale=False):
        captured['labels_mask'] = [t.clone() for t in dl]
        rh.mask_roi_extractor = lambda f, rois: feats
        rh.mask_roi_extractor.num_inputs = 1
        rh.shared_head = lambda t: t
        rh.mask_head = lambda t: t
        rh.share_roi_extractor = False
        res = rh._mask_forward(torch.zeros(1, 1, 1), rois=torch.zeros(6, 5))
        captured['mask_pred'] = res['mask_pred']
        return 'segm'
    rh.simple_test_mask = fake_simple_test_mask
    rh.test_cfg = None
    out = rh.simple_test(None, None, None)
    assert out[2] == 'segm'
    np.savez_compressed(os.path.join(OUT, 'mask_gather.npz'), cat_mean_mp=mp.numpy(),
                        det_labels_0=det_labels[0].numpy(), det_labels_1=det_labels[1].numpy(),
                        feats=feats.numpy(), spp_vecs_mask=rh.spp_vecs_mask.numpy(),
                        mask_pred=captured['mask_pred'].numpy(),
                        labels_mask_0=captured['labels_mask'][0].numpy())
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
