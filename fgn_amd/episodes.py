"""Seeded synthetic few-shot episodes in the FewShotISEG batch layout.

Produces exactly the dict ``collate_fn_new`` builds from ``BaseFewShotISEG``
samples (reference: subprojects/sp02_omniiseg_fgn_mmdet/main.py:62-76 and
datasets/fewshotiseg/base_fst.py:1248-1266): stacked ``qry_img``, ``spp_imgs``,
``spp_bboxes``, ``spp_isegmaps``, ``img_shape`` plus per-image lists for the
ragged query annotations.  Boxes are YXYX as in the dataset (README.md:71).
There are no COCO/VOC images in the build container, so pixel content is
synthetic: post-normalisation-scale noise with a few textured objects.
"""
from __future__ import annotations

import numpy as np
import torch


def _ellipse_mask(h, w, y0, x0, y1, x1):
    yy = (np.arange(h, dtype=np.float32) + 0.5)[:, None]
    xx = (np.arange(w, dtype=np.float32) + 0.5)[None, :]
    cy, cx = (y0 + y1) / 2.0, (x0 + x1) / 2.0
    ry, rx = max((y1 - y0) / 2.0, 1.0), max((x1 - x0) / 2.0, 1.0)
    return (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1.0


def make_episode(idx: int, n_ways: int, k_shots: int, height: int, width: int,
                 spp_size: int, n_qry_objs: int = 4, seed: int = 1234,
                 spp_fill_ratio: float = 0.8) -> dict:
    """One sample dict (un-collated) for episode ``idx``."""
    rng = np.random.RandomState(seed + idx)
    g = torch.Generator().manual_seed(seed + idx)
    qry = torch.randn(3, height, width, generator=g)
    class_tint = rng.randn(n_ways, 3).astype(np.float32)

    bboxes, cat_ids, masks = [], [], []
    for j in range(n_qry_objs):
        bh = rng.randint(max(height // 8, 8), max(height // 2, 9))
        bw = rng.randint(max(width // 8, 8), max(width // 2, 9))
        y0 = rng.randint(0, height - bh)
        x0 = rng.randint(0, width - bw)
        c = int(rng.randint(0, n_ways))
        m = _ellipse_mask(height, width, y0, x0, y0 + bh, x0 + bw)
        qry += torch.from_numpy(m.astype(np.float32))[None] * \
            torch.from_numpy(class_tint[c])[:, None, None] * 1.5
        bboxes.append([y0, x0, y0 + bh, x0 + bw])
        cat_ids.append(c)
        masks.append(m)

    nk = n_ways * k_shots
    spp = torch.randn(nk, 3, spp_size, spp_size, generator=g)
    spp_boxes = np.zeros((nk, 4), np.float32)
    spp_masks = np.zeros((nk, spp_size, spp_size), bool)
    for n in range(n_ways):
        for k in range(k_shots):
            i = n * k_shots + k          # class-major (base_fst.py:1054-1080)
            side = spp_size * spp_fill_ratio * (0.8 + 0.2 * rng.rand())
            off_y = (spp_size - side) / 2 + rng.uniform(-2, 2)
            off_x = (spp_size - side) / 2 + rng.uniform(-2, 2)
            box = [off_y, off_x, off_y + side, off_x + side * (0.7 + 0.3 * rng.rand())]
            spp_boxes[i] = box
            m = _ellipse_mask(spp_size, spp_size, *box)
            spp_masks[i] = m
            spp[i] += torch.from_numpy(m.astype(np.float32))[None] * \
                torch.from_numpy(class_tint[n])[:, None, None] * 1.5

    return {
        'idx': idx,
        'qry_child_idx': idx,
        'qry_img': qry,
        'qry_cat_ids': np.asarray(cat_ids, np.int64),
        'qry_bboxes': np.asarray(bboxes, np.float32).reshape(-1, 4),
        'qry_isegmaps': np.stack(masks).astype(bool) if masks else
        np.zeros((0, height, width), bool),
        'spp_imgs': spp,
        'spp_bboxes': spp_boxes,
        'spp_isegmaps': spp_masks,
        'cats_ids_to_sample_real': np.arange(n_ways, dtype=np.int64) + 1,
        'spp_insts_ids': np.arange(nk, dtype=np.int64) + 100 * idx,
        'img_shape': np.array([height, width, 3], dtype=np.int32),
    }


_LIST_KEYS = ('qry_cat_ids_real', 'qry_cat_ids', 'qry_bboxes', 'qry_isegmaps')


def collate(samples: list) -> dict:
    """Restatement of ``collate_fn_new`` (main.py:62-76): ragged query
    annotations stay lists of tensors, everything else is stacked."""
    batch = {}
    for key in samples[0]:                       # default_collate of everything that stacks, in sample-dict order ...
        if key in _LIST_KEYS:
            continue
        vals = [s[key] for s in samples]
        if isinstance(vals[0], torch.Tensor):
            batch[key] = torch.stack(vals)
        else:
            batch[key] = torch.as_tensor(np.stack([np.asarray(v) for v in vals]))
    for key in _LIST_KEYS:                       # ... then the ragged keys as lists of tensors, in the reference's order
        if key in samples[0]:
            batch[key] = [torch.as_tensor(s[key]) for s in samples]
    return batch


def make_batch(first_idx: int, batch: int, n_ways: int, k_shots: int,
               height: int, width: int, spp_size: int, **kw) -> dict:
    return collate([make_episode(first_idx + i, n_ways, k_shots, height, width,
                                 spp_size, **kw) for i in range(batch)])


# the BASELINE.json configurations
CONFIGS = {
    'cfg1': dict(n_ways=1, k_shots=1, height=128, width=128, spp_size=128),
    'cfg2': dict(n_ways=3, k_shots=1, height=256, width=256, spp_size=128),
    'cfg3': dict(n_ways=3, k_shots=3, height=800, width=1333, spp_size=256),
    'cfg4': dict(n_ways=3, k_shots=3, height=800, width=1328, spp_size=256),
    # 5-way 5-shot, 1000 proposals: a build extension, the reference asserts N in {1,3}
    'cfg5': dict(n_ways=5, k_shots=5, height=1024, width=1024, spp_size=256),
}
RPN_MAX_PER_IMG = {'cfg5': 1000}
