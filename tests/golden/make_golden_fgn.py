#!/usr/bin/env python3
"""Golden vectors for the glue of the reference's detector file ITSELF:
subprojects/sp02_omniiseg_fgn_mmdet/fgn.py -- ``FGN.modify_input`` (79-108), ``FGN.get_img_metas``
(110-123) and the orchestration + result-packing loop of ``FGN.simple_test`` (187-303).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_fgn.py

fgn.py is imported unmodified with the absent third-party packages replaced by empty module objects
(the stubs of make_golden.py plus cv2 / imgaug / the reference's own ``datasets`` package, which the
installed HuggingFace ``datasets`` distribution shadows).  ``FGN`` is instantiated through ``__new__``
(its constructor builds mmdet modules); backbone, AG-RPN head and RoI head are replaced by recorders that
return fixed seeded tensors, so everything BETWEEN them -- the reference's own lines -- runs as written:
  * YXYX -> XYXY swap of ``qry_bboxes[i]`` / ``spp_bboxes[i]``, which objects it mutates;
  * the views handed to the heads (``spp_imgs.view(-1,c,h,w)``, ``spp_bboxes.view(-1,1,4)``,
    ``spp_isegmaps.view(-1,1,h,w)``), metas (``img_shape / ori_shape / pad_shape``, ``scale_factor`` ones);
  * ``proposal_cfg = test_cfg.get('rpn_proposal', test_cfg.rpn)``;
  * packing: ``outputs_all[2][i][0]``, score column, ``[1,0,3,2]`` reorder, passthrough keys, tensor -> numpy,
    ``qry_isegmaps`` popped and replaced by ``qry_isegmaps_rle``.
Not pinned (stubs): ``encode_mask_results`` (mmdet -> pycocotools) is the oracle's RLE encoder; ``Tensor.cuda()``
is a clone (no GPU in the build container).  Two runs: ``define_env()`` as found here ('OTHER': no device
copies, the CALLER's ``qry_bboxes`` list is swapped in place and passed through as XYXY) and forced to
'SERVER' (the environment of the published numbers: ``.cuda()`` copies shield the caller, passthrough stays
YXYX) -- the build follows SERVER.
Nothing from the reference is copied: this script imports it, feeds seeded inputs and stores outputs.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)


class _Cfg(dict):
    """mmcv ConfigDict stand-in: attribute access on a dict."""
    __getattr__ = dict.__getitem__


def _import_reference_fgn():
    import make_golden as mg
    mg._stub_third_party()

    def mod(name, **attrs):
        m = types.ModuleType(name)
        sys.modules[name] = m
        for k, v in attrs.items():
            setattr(m, k, v)
        return m
    for k in [k for k in sys.modules if k == 'datasets' or k.startswith('datasets.')]:
        del sys.modules[k]
    mod('datasets'); mod('datasets.fewshotiseg'); mod('datasets.fewshotiseg.base_fst', BaseFewShotISEG=object)
    mod('cv2')
    mod('imgaug'); mod('imgaug.augmenters', Resize=None, Pad=None)
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mod('matplotlib'); mod('matplotlib.pyplot')
    from oracle import fgn_ref_cpu as O
    sys.modules['mmdet.core'].encode_mask_results = \
        lambda cls_segms: [[O.rle_encode(np.asarray(m)) for m in c] for c in cls_segms]
    sys.path.insert(0, REF)
    from subprojects.sp02_omniiseg_fgn_mmdet import fgn as ref_fgn
    return ref_fgn


def make_inputs(seed=20261004):
    g = torch.Generator().manual_seed(seed)
    B, N, K, S, H, W = 2, 3, 2, 16, 24, 40
    n_gt = [3, 1]
    box = lambda n, h, w: torch.stack([torch.rand(n, generator=g) * h * .4, torch.rand(n, generator=g) * w * .4,
                                       h * .5 + torch.rand(n, generator=g) * h * .5,
                                       w * .5 + torch.rand(n, generator=g) * w * .5], 1)      # YXYX
    return dict(
        qry_img=torch.randn(B, 3, H, W, generator=g),
        qry_bboxes=[box(n, H, W) for n in n_gt],
        qry_cat_ids=[torch.randint(0, N, (n,), generator=g) for n in n_gt],
        qry_isegmaps=[torch.rand(n, H, W, generator=g) > 0.6 for n in n_gt],
        spp_imgs=torch.randn(B, N * K, 3, S, S, generator=g),
        spp_bboxes=torch.stack([box(N * K, S, S) for _ in range(B)]),
        spp_isegmaps=torch.rand(B, N * K, S, S, generator=g) > 0.5,
        qry_child_idx=[torch.tensor([5, 2, 9]), torch.tensor([1])],
        img_shape=torch.tensor([[H, W, 3], [H, W, 3]], dtype=torch.int32),
        cats_ids_to_sample_real=[torch.tensor([7, 11, 3]), torch.tensor([2, 5, 19])],
        spp_insts_ids=[torch.arange(N * K) + 100, torch.arange(N * K) + 200],
        idx=torch.tensor([41, 42]))


def make_head_outputs(seed=77):
    """What the stubbed heads return: fixed detections / masks (mmdet result layout)."""
    g = torch.Generator().manual_seed(seed)
    H, W = 24, 40
    n_det = [4, 0]
    det = [torch.cat([torch.rand(n, 4, generator=g) * 20, torch.rand(n, 1, generator=g)], 1) for n in n_det]
    lab = [torch.randint(0, 3, (n,), generator=g) for n in n_det]
    seg = [[[(torch.rand(H, W, generator=g) > 0.5).numpy() for _ in range(n)]] for n in n_det]
    return det, lab, seg


def run(ref_fgn, env):
    rec = {}
    ins = make_inputs()
    caller = {k: ([t.clone() for t in v] if isinstance(v, list) else v.clone()) for k, v in ins.items()}
    det, lab, seg = make_head_outputs()
    g = torch.Generator().manual_seed(5)
    B = ins['qry_img'].shape[0]
    qf = torch.randn(B, 8, 2, 3, generator=g)
    sf = torch.randn(B * 6, 8, 1, 1, generator=g)

    model = ref_fgn.FGN.__new__(ref_fgn.FGN)
    nn.Module.__init__(model)
    model.n_ways, model.k_shots = 3, 2
    model.test_cfg = _Cfg(rpn=_Cfg(nms_pre=6000, tag='rpn'), rcnn=_Cfg(tag='rcnn'))

    def extract_feat(img):
        rec.setdefault('extract_shapes', []).append(tuple(img.shape))
        rec.setdefault('extract_sums', []).append(float(img.double().sum()))
        return (qf,) if len(rec['extract_shapes']) == 1 else (sf,)
    model.extract_feat = extract_feat

    class RPN:
        def forward_single(self, q, s):
            rec['rpn_in_is_fmaps'] = (q is qf) and (s is sf)
            return 'CLS', 'REG'

        def get_bboxes(self, cls, reg, img_metas=None, cfg=None):
            rec['get_bboxes_args'] = (cls, reg)
            rec['metas'] = img_metas
            rec['proposal_cfg_tag'] = cfg['tag']
            return ['P0', 'P1']

    class ROI:
        def simple_test(self, x, proposal_list, img_metas, rescale=False, spp_fmaps=None, spp_bboxes=None,
                        spp_isegmaps=None):
            rec['roi_args'] = dict(x_is_qf=x is qf, proposals=proposal_list, metas_same=img_metas is rec['metas'],
                                   rescale=rescale, spp_fmaps_is_sf=spp_fmaps is sf,
                                   spp_bboxes=spp_bboxes.clone(), spp_isegmaps=spp_isegmaps.clone())
            return det, lab, seg
    object.__setattr__(model, 'rpn_head', RPN())
    object.__setattr__(model, 'roi_head', ROI())

    saved_env, saved_cuda = ref_fgn.define_env, torch.Tensor.cuda
    ref_fgn.define_env = lambda: env
    torch.Tensor.cuda = lambda self, *a, **k: self.clone()       # a device copy is a new tensor
    try:
        out = model.simple_test(**caller, rescale=True)
    finally:
        ref_fgn.define_env, torch.Tensor.cuda = saved_env, saved_cuda
    return ins, caller, rec, out, (det, lab, seg)


def flatten(prefix, ins, caller, rec, out, heads, store):
    det, lab, seg = heads
    for k, v in caller.items():           # the caller's objects AFTER the call (mutation check)
        if isinstance(v, list):
            for i, t in enumerate(v):
                store[f'{prefix}caller_after__{k}__{i}'] = t.numpy()
        else:
            store[f'{prefix}caller_after__{k}'] = v.numpy()
    store[prefix + 'extract_shapes'] = np.array([list(s) + [0] * (4 - len(s)) for s in rec['extract_shapes']])
    store[prefix + 'extract_sums'] = np.array(rec['extract_sums'])
    store[prefix + 'wiring_ok'] = np.array([rec['rpn_in_is_fmaps'], rec['get_bboxes_args'] == (['CLS'], ['REG']),
                                            rec['roi_args']['x_is_qf'], rec['roi_args']['metas_same'],
                                            rec['roi_args']['spp_fmaps_is_sf'], rec['roi_args']['rescale'] is True,
                                            rec['roi_args']['proposals'] == ['P0', 'P1'],
                                            rec['proposal_cfg_tag'] == 'rpn'])
    store[prefix + 'roi_spp_bboxes'] = rec['roi_args']['spp_bboxes'].numpy()
    store[prefix + 'roi_spp_isegmaps'] = rec['roi_args']['spp_isegmaps'].numpy()
    for i, m in enumerate(rec['metas']):
        assert sorted(m) == ['img_shape', 'num_samples', 'ori_shape', 'pad_shape', 'scale_factor'], sorted(m)
        for k in ('img_shape', 'ori_shape', 'pad_shape'):
            assert m[k].device.type == 'cpu' and m[k].dtype == torch.int32
            store[f'{prefix}meta{i}__{k}'] = m[k].numpy()
        store[f'{prefix}meta{i}__scale_factor'] = m['scale_factor']
        assert m['num_samples'] == 1
    store[prefix + 'n_out'] = np.array(len(out))
    keys = None
    for i, o in enumerate(out):
        assert keys is None or keys == list(o), 'key order differs between images'
        keys = list(o)
        for k, v in o.items():
            if k.endswith('_rle'):
                store[f'{prefix}out{i}__{k}__n'] = np.array(len(v))
                for j, r in enumerate(v):
                    store[f'{prefix}out{i}__{k}__{j}__size'] = np.array(r['size'])
                    store[f'{prefix}out{i}__{k}__{j}__counts'] = np.frombuffer(r['counts'], np.uint8)
            else:
                assert isinstance(v, np.ndarray), (k, type(v))
                store[f'{prefix}out{i}__{k}'] = v
    store[prefix + 'out_keys'] = np.array(keys)
    for i in range(len(det)):
        store[f'{prefix}head_det{i}'] = det[i].numpy()
        store[f'{prefix}head_lab{i}'] = lab[i].numpy()
        store[f'{prefix}head_seg{i}'] = np.stack(seg[i][0]) if seg[i][0] else np.zeros((0, 24, 40), bool)


def main():
    ref_fgn = _import_reference_fgn()
    assert ref_fgn.define_env() == 'OTHER'
    store = {}
    for env, prefix in (('SERVER', 'server__'), ('OTHER', 'other__')):
        flatten(prefix, *run(ref_fgn, env), store)
    np.savez_compressed(os.path.join(HERE, 'fgn_glue.npz'), **store)
    print('written', os.path.join(HERE, 'fgn_glue.npz'), len(store), 'arrays')


if __name__ == '__main__':
    main()
