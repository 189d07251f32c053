#!/usr/bin/env python3
"""Golden vectors for the episode producer from the reference's own data-side files:

  cp_utils/create_img_from_chars.py:250-267      get_new_shape (the 800 / 1333 resize rule)
  datasets/fewshotiseg/base_fst.py:991-1040      BaseFewShotISEG.cut_algorithm, .get_crop (support crop geometry:
                                                 offsets, squaring, reflect / constant padding, box in the crop)
  datasets/fewshotiseg/base_fst.py:605-732       BaseFewShotISEG.reshuffle, aspect-ratio-grouped branch (groups by
                                                 rounded w/h, per-group size via get_new_shape rounded to x16,
                                                 padding of groups by re-drawing members, chunking, chunk shuffle)
  subprojects/sp02_omniiseg_fgn_mmdet/main.py:62-76   collate_fn_new (the batch dict the hot path receives: stacked
                                                 tensors first, the four ragged query keys as lists of tensors last)

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_data.py

Both files are imported unmodified; the third-party packages they import and this image lacks (cv2, imgaug,
imagesize, printy, torchvision, pycocotools) are replaced by permissive stand-in modules - none of them is reached
by the functions exercised here, except ``imagesize.get``, which the script answers from a table of synthetic image
sizes (the reference reads them from image files).  ``reshuffle`` runs on an instance made with ``__new__`` whose
attributes are set by hand; ``random`` is seeded so that the re-draws and shuffles can be replayed.
NOT pinned (imgaug, third party): ``iaa.Resize`` / ``iaa.CenterPadToFixedSize`` of get_support (base_fst.py:478-482).
Nothing from the reference is copied: this script imports it, feeds seeded inputs and stores outputs.
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


class _Any:
    """Permissive stand-in: any call / attribute / subscript gives another stand-in."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Any()

    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        return _Any()

    def __getitem__(self, k):
        return _Any()

    def __iter__(self):
        return iter(())


class _AnyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__') and name.endswith('__'):
            raise AttributeError(name)
        if name[:1].isupper():
            return type(name, (_Any,), {})
        return _Any()


def import_reference():
    # the reference pins numpy 1.19 (requirements.txt), where `np.str` / `np.bool` still alias the builtins
    # (base_fst.py:702, 1164); numpy 2.2 of this image removed the aliases
    for alias, typ in (('str', str), ('bool', bool)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    for name in ('cv2', 'imgaug', 'imgaug.augmenters', 'imagesize', 'printy', 'torchvision', 'torchvision.ops',
                 'torchvision.transforms', 'pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval', 'pycocotools.coco',
                 'mmdet', 'mmdet.core'):
        sys.modules[name] = _AnyModule(name)
    sys.modules['pycocotools.cocoeval'].COCOeval = type('COCOeval', (), {})
    # the reference's top-level `datasets` directory is shadowed by the installed HuggingFace distribution
    for k in [k for k in sys.modules if k == 'datasets' or k.startswith('datasets.')]:
        del sys.modules[k]
    for name, path in (('datasets', 'datasets'), ('datasets.fewshotiseg', 'datasets/fewshotiseg')):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m
    sys.path.insert(0, REF)
    from cp_utils import create_img_from_chars as cic
    from datasets.fewshotiseg import base_fst
    return cic, base_fst


# ---- seeded inputs (importable without the reference: the tests rebuild them) -------------------------------
def new_shape_cases():
    return [(h, w) for h in list(range(60, 1500, 97)) + [375, 480, 333, 500, 640, 800, 1333]
            for w in list(range(70, 1500, 89)) + [500, 640, 1000, 480, 800, 1333]]


def offset_ratio(fill: float) -> float:
    return float(np.around(1 / (2 * fill) - 0.5, decimals=2))         # base_fst.py:264-265


def crop_cases():
    """(image [H,W,3] u8, ymin, xmin, ymax, xmax, h_offset, w_offset, crop_square, mode)"""
    rng = np.random.RandomState(20261004)
    cases = []
    for k in range(18):
        H, W = int(rng.randint(24, 48)), int(rng.randint(24, 48))
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        ymin, xmin = int(rng.randint(0, H // 2)), int(rng.randint(0, W // 2))
        ymax, xmax = int(rng.randint(ymin + 3, H + 1)), int(rng.randint(xmin + 3, W + 1))
        fill = (0.8, 0.8, 1.0, 0.5, 0.65, 0.9)[k % 6]
        r = offset_ratio(fill)
        h_off, w_off = int(np.floor((ymax - ymin) * r)), int(np.floor((xmax - xmin) * r))     # base_fst.py:1112-1113
        cases.append((img, ymin, xmin, ymax, xmax, h_off, w_off, bool(k % 5 != 4), 'reflect' if k % 2 == 0 else 'constant'))
    return cases


def cut_cases():
    rng = np.random.RandomState(7)
    return [(int(a), int(a + d), int(o), int(a + d + e)) for a, d, o, e in
            zip(rng.randint(0, 50, 40), rng.randint(1, 60, 40), rng.randint(0, 30, 40), rng.randint(0, 40, 40))]


def ar_sizes(n=57, seed=3):
    """(width, height) of n synthetic dataset images: COCO-like aspect ratios."""
    rng = np.random.RandomState(seed)
    pool = [(640, 480), (480, 640), (500, 375), (640, 427), (427, 640), (500, 333), (640, 640), (612, 612), (640, 360),
            (333, 500), (500, 400), (640, 512)]
    sizes = [pool[i] for i in rng.randint(0, len(pool), n)]
    sizes[5] = (1000, 300)          # one very wide image: a group of its own, long side capped at max_size
    return sizes


def collate_samples():
    """Two sample dicts shaped like ``BaseFewShotISEG.__getitem__`` yields them (base_fst.py:1248-1266): tensors for the
    images, numpy arrays / ints for the rest, ragged query annotations."""
    import torch
    g = torch.Generator().manual_seed(77)
    rng = np.random.RandomState(77)
    out = []
    for i in range(2):
        n = 1 + 2 * i
        out.append({'idx': 40 + i, 'qry_child_idx': 3 + i, 'qry_img': torch.randn(3, 8, 10, generator=g),
                    'qry_cat_ids_real': rng.randint(0, 20, n), 'qry_cat_ids': rng.randint(0, 3, n),
                    'qry_bboxes': rng.rand(n, 4).astype(np.float32), 'qry_isegmaps': rng.rand(n, 8, 10) > 0.5,
                    'spp_imgs': torch.randn(6, 3, 4, 4, generator=g), 'spp_bboxes': rng.rand(6, 4).astype(np.float32),
                    'spp_isegmaps': rng.rand(6, 4, 4) > 0.5, 'cats_ids_to_sample_real': rng.randint(0, 20, 3),
                    'cats_ids_to_sample': np.arange(3), 'spp_insts_ids': rng.randint(0, 99, 6),
                    'img_shape': np.array([8, 10, 3], dtype=np.int32)})
    return out


def import_reference_main():
    """subprojects/sp02_omniiseg_fgn_mmdet/main.py for its ``collate_fn_new`` (main.py:62-76): mmcv / mmdet and the
    dataset / detector modules it imports at the top are stand-ins (none is reached by the function)."""
    for name in ('mmcv', 'mmcv.runner', 'mmcv.runner.hooks', 'mmcv.runner.hooks.logger', 'mmcv.runner.base_module',
                 'mmdet.utils', 'mmdet.models', 'datasets.fewshotiseg.mnistiseg_fst', 'datasets.fewshotiseg.omniiseg_fst',
                 'datasets.fewshotiseg.coco_fst', 'datasets.fewshotiseg.voc_fst', 'subprojects.sp02_omniiseg_fgn_mmdet.fgn'):
        sys.modules[name] = _AnyModule(name)
    from subprojects.sp02_omniiseg_fgn_mmdet import main as ref_main
    return ref_main


def main():
    cic, base_fst = import_reference()
    store = {}
    # ---- collate_fn_new (main.py:62-76)
    import torch
    ref_main = import_reference_main()
    batch = ref_main.collate_fn_new(collate_samples())
    store['collate_keys'] = np.array(list(batch))
    for k, v in batch.items():
        if isinstance(v, list):
            store[f'collate__{k}__n'] = np.array(len(v))
            for j, t in enumerate(v):
                assert isinstance(t, torch.Tensor)
                store[f'collate__{k}__{j}'] = t.numpy()
        else:
            assert isinstance(v, torch.Tensor)
            store[f'collate__{k}'] = v.numpy()
    # ---- get_new_shape
    out = []
    for h, w in new_shape_cases():
        try:
            out.append(np.asarray(cic.get_new_shape(h, w, 800, 1333)).astype(np.int64))
        except AssertionError:      # the reference asserts |AR_old - AR_new| <= 0.015
            out.append(np.array([-1, -1]))
    store['new_shape_hw'] = np.array(new_shape_cases())
    store['new_shape_out'] = np.stack(out)
    store['new_shape_small'] = np.stack([np.asarray(cic.get_new_shape(h, w, 128, 256)) for h, w in ((100, 100), (64, 200), (300, 100))])
    # ---- cut_algorithm / get_crop
    B = base_fst.BaseFewShotISEG
    store['cut_in'] = np.array(cut_cases())
    store['cut_out'] = np.stack([B.cut_algorithm(a, b, o, m) for a, b, o, m in cut_cases()])
    for i, (img, ymin, xmin, ymax, xmax, ho, wo, sq, mode) in enumerate(crop_cases()):
        crop, box = B.get_crop(img, ymin, xmin, ymax, xmax, ho, wo, crop_square=sq, mode=mode)
        store[f'crop{i}_out'], store[f'crop{i}_box'] = crop, box
        m2 = (img[..., :1] > 127)
        mcrop, _ = B.get_crop(m2, ymin, xmin, ymax, xmax, ho, wo, crop_square=sq, mode='constant')    # the mask call
        store[f'crop{i}_mask'] = mcrop
    store['crop_n'] = np.array(len(crop_cases()))
    # ---- reshuffle, aspect-ratio-grouped branch
    sizes = ar_sizes()
    paths = [f'img_{i:04d}.jpg' for i in range(len(sizes))]
    sys.modules['imagesize'].get = lambda p: sizes[paths.index(p)]
    base_fst.imagesize = sys.modules['imagesize']
    for tag, shuffle, batch, seed in (('a', True, 4, 11), ('b', False, 4, 12), ('c', True, 3, 13)):
        ds = B.__new__(B)
        ds.batch, ds.sampling_origin_ds, ds.merged_ds, ds.shuffle = batch, 'COCO', None, shuffle
        ds.order_initial = np.arange(len(sizes), dtype=np.int32)
        ds.target_size, ds.max_size, ds.sub_sample_ratio = 800, 1333, 16
        ds.a_print = lambda *a, **k: None
        ds.__getitem__ = lambda idx, path_only=False: paths[int(idx)]
        random.seed(seed)
        ds.reshuffle()
        store[f'ar_{tag}_order'] = np.asarray(ds.order)
        store[f'ar_{tag}_groups'] = np.asarray(ds.ar_groups_group_indexes_all)
        store[f'ar_{tag}_hws'] = np.asarray(ds.ar_group_new_hws)
        store[f'ar_{tag}_cfg'] = np.array([int(shuffle), batch, seed])
    np.savez_compressed(os.path.join(HERE, 'data_side.npz'), **store)
    print('written', os.path.join(HERE, 'data_side.npz'), len(store), 'arrays')


if __name__ == '__main__':
    main()
