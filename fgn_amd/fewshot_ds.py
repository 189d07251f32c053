"""Synthetic episodic dataset with the BaseFewShotISEG interface.

The reference's datasets (datasets/fewshotiseg/base_fst.py) need cv2/imgaug and real
images; the hot path only depends on their *sample dict* (base_fst.py:1248-1266) and on
``evaluate`` (base_fst.py:1516-1601).  This class produces that contract from seeded
synthetic episodes so the reference's evaluation loop (main.py:269-326) runs unchanged
on top of ``fgn_amd.detector.FGN``; visualisation side effects of ``evaluate`` are not
reproduced.
"""
from __future__ import annotations

import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from . import cluttered_chars as cc
from . import episodes
from .fsiseg_eval import FSISEGEval


def get_new_shape(h, w, target_size=800, max_size=1333, check_ar: bool = True):
    """Short side -> target_size, long side = target_size * aspect ratio (truncated), capped at max_size with the
    short side re-derived (cp_utils/create_img_from_chars.py:250-267).  Like the reference, raises AssertionError
    when the truncations move the aspect ratio by more than 0.015 (tiny inputs).  Pinned by
    tests/golden/data_side.npz (the reference's own function on 506 shapes)."""
    old = np.array([h, w], dtype=np.float64)
    new = np.array([h, w], dtype=np.int64)
    long_i = int(np.argmax(np.array([h, w])))
    ar = old[long_i] / old[1 - long_i]
    new[1 - long_i] = target_size
    new[long_i] = int(target_size * ar)
    if new[long_i] > max_size:
        new[long_i] = max_size
        new[1 - long_i] = int(max_size / ar)
    if check_ar:
        assert abs(old[0] / old[1] - new[0] / float(new[1])) <= 0.015, f'Old {old[0] / old[1]} New {new[0] / float(new[1])}'
    return new


def ar_grouped_order(aspect_ratios, batch, target_size=800, max_size=1333, sub_sample_ratio=16, shuffle=True,
                     seed=0, rnd=None):
    """Aspect-ratio-grouped batching (``BaseFewShotISEG.reshuffle``, base_fst.py:626-727): width/height rounded to
    one decimal, one group per value; per-group target (h, w) = get_new_shape(100 / ar, 100) rounded to multiples of
    ``sub_sample_ratio``; every group is padded to a multiple of ``batch`` by re-drawing its own members
    (``random.choices``, whether or not ``shuffle``), shuffled, cut into chunks of ``batch``; the chunks are shuffled.
    Returns (order [n_chunks*batch], group index per entry, per-group (h, w)).  The reference draws from the global
    ``random`` module; here from ``rnd`` (default ``random.Random(seed)``) in the same call order, so
    ``random.seed(s)`` there and ``seed=s`` here give the same order (tests/golden/data_side.npz)."""
    rnd = random.Random(seed) if rnd is None else rnd
    ars = np.around(np.asarray(aspect_ratios, np.float64), decimals=1)
    uniq = sorted(np.unique(ars))
    hws = []
    for a in uniq:
        hws.append(get_new_shape(100 / a, 100, target_size, max_size))
    hws = (np.around(np.array(hws) / sub_sample_ratio) * sub_sample_ratio).astype(np.int32).reshape(-1, 2)
    order, groups = [], []
    for gi, a in enumerate(uniq):
        members = np.flatnonzero(ars == a)
        elems = list(range(len(members)))
        if len(members) % batch:
            elems += rnd.choices(elems, k=batch - len(members) % batch)
        if shuffle:
            rnd.shuffle(elems)
        order.extend(members[elems].tolist())
        groups.extend([gi] * len(elems))
    order = np.asarray(order, np.int32).reshape(-1, batch)
    groups = np.asarray(groups, np.int32).reshape(-1, batch)
    chunks = list(range(len(order)))
    if shuffle:
        rnd.shuffle(chunks)
    return order[chunks].reshape(-1), groups[chunks].reshape(-1), hws


# ---- support crop geometry (BaseFewShotISEG.get_support, base_fst.py:1043-1167) ------------------------------------
def spp_offset_ratio(spp_fill_ratio: float) -> float:
    """base_fst.py:264-265: 1 / (1 + 2 * offset) = fill ratio, rounded to two decimals (0.8 -> 0.12)."""
    return float(np.around(1 / (2 * spp_fill_ratio) - 0.5, decimals=2))


def cut_algorithm(lo: int, hi: int, offset: int, max_shape: int) -> np.ndarray:
    """``BaseFewShotISEG.cut_algorithm`` (base_fst.py:991-998): cut window [lo - offset, hi + offset] clipped to
    [0, max_shape] and the object's extent inside it -> int32 [cut_lo, box_lo, cut_hi, box_hi]."""
    cut_lo = max(0, lo - offset)
    return np.array([cut_lo, lo - cut_lo, min(hi + offset, max_shape), hi - cut_lo], dtype=np.int32)


def get_crop(img: np.ndarray, ymin, xmin, ymax, xmax, h_offset, w_offset, crop_square=True, mode='reflect'):
    """``BaseFewShotISEG.get_crop`` (base_fst.py:1000-1040; pinned by tests/golden/data_side.npz): the crop of an
    instance with ``offset`` pixels of context on every side.  Context beyond the image border comes from np.pad of
    the WHOLE image at its top/left (pad = the offsets; bottom/right one more when the padded extent is odd), in
    ``mode`` ('reflect' for the image, 'constant' for the mask); with ``crop_square`` the shorter side's offset grows
    until the window is square; without it the padding is constant zeros.  Zero offsets: only the parity padding.
    img [H,W,C] -> (crop, box int32 YXYX of the instance inside the crop)."""
    h, w = ymax - ymin, xmax - xmin
    if h_offset == 0 and w_offset == 0:
        pads = ((0, h % 2), (0, w % 2))
        box = (0, 0, h, w)
    else:
        if crop_square:
            eh, ew = h + 2 * h_offset, w + 2 * w_offset
            eh, ew = eh + eh % 2, ew + ew % 2
            if eh > ew:
                w_offset += (eh - ew) // 2
            elif ew > eh:
                h_offset += (ew - eh) // 2
        else:
            mode = 'constant'
        pads = ((h_offset, h_offset + (h + 2 * h_offset) % 2), (w_offset, w_offset + (w + 2 * w_offset) % 2))
        box = (h_offset, w_offset, h_offset + h, w_offset + w)
    padded = np.pad(img, [list(pads[0]), list(pads[1]), [0, 0]], mode=mode)
    crop = padded[ymin:ymax + sum(pads[0]), xmin:xmax + sum(pads[1])]
    return crop, np.array(box, dtype=np.int32).reshape(4)


def _resize(img: np.ndarray, nh: int, nw: int, binary: bool) -> np.ndarray:
    """imgaug ``Resize`` stand-in (third party, not pinned): bicubic like its default interpolation; a mask is
    resized as float and re-thresholded at 0.5 (imgaug's handling of bool arrays)."""
    import torch.nn.functional as F
    t = torch.from_numpy(np.ascontiguousarray(img)).float()
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    t = F.interpolate(t, size=(nh, nw), mode='bicubic', align_corners=False)
    if binary:
        return (t[0, 0] > 0.5).numpy()
    return t[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()


def support_from_instance(img: np.ndarray, bbox_yxyx, isegmap: np.ndarray, spp_img_size: int,
                          spp_fill_ratio: float = 0.8, crop_square: bool = True):
    """One support sample as ``BaseFewShotISEG.get_support`` builds it (base_fst.py:1104-1155): integer box ->
    offsets ``floor(extent * offset_ratio)`` -> ``get_crop`` (image: reflect context, mask: zeros) ->
    ``iaa.Resize({'longer-side': S, 'shorter-side': 'keep-aspect-ratio'})`` -> ``iaa.CenterPadToFixedSize(S, S,
    pad_mode='constant')`` (base_fst.py:478-482) with the box carried along.  The crop geometry is pinned by the
    reference's golden; the two imgaug operators are third-party and restated (bicubic resize, shorter side =
    round(S * short / long), centre padding with the odd pixel on the bottom / right).
    img [H,W,3] uint8, isegmap [H,W] bool -> (crop [S,S,3] uint8, box float32 YXYX in crop pixels, mask [S,S] bool)."""
    ymin, xmin, ymax, xmax = np.array(bbox_yxyx).astype(np.int32)
    r = spp_offset_ratio(spp_fill_ratio)
    w_off, h_off = int(np.floor((xmax - xmin) * r)), int(np.floor((ymax - ymin) * r))
    crop, box = get_crop(img, ymin, xmin, ymax, xmax, h_off, w_off, crop_square=crop_square)
    mcrop, _ = get_crop(isegmap[:, :, None], ymin, xmin, ymax, xmax, h_off, w_off, crop_square=crop_square,
                        mode='constant')
    ch, cw = crop.shape[:2]
    S = int(spp_img_size)
    if ch >= cw:
        nh, nw = S, max(1, int(np.round(S * cw / ch)))
    else:
        nh, nw = max(1, int(np.round(S * ch / cw))), S
    crop_r, mask_r = _resize(crop, nh, nw, False), _resize(mcrop[:, :, 0], nh, nw, True)
    box_r = box.astype(np.float32) * np.array([nh / ch, nw / cw, nh / ch, nw / cw], np.float32)
    top, left = (S - nh) // 2, (S - nw) // 2
    out = np.zeros((S, S, 3), np.uint8)
    out[top:top + nh, left:left + nw] = crop_r
    m = np.zeros((S, S), bool)
    m[top:top + nh, left:left + nw] = mask_r
    return out, (box_r + np.array([top, left, top, left], np.float32)).astype(np.float32), m


class SyntheticFewShotISEG(Dataset):
    def __init__(self, n_ways=3, k_shots=3, length=64, height=800, width=1333, spp_img_size=256, batch=4,
                 seed=1234, suffix='SYNTH_val_novel'):
        self.n_ways, self.k_shots = n_ways, k_shots
        self.length, self.batch = length, batch
        self.height, self.width, self.spp_img_size = height, width, spp_img_size
        if batch > 1:       # the reference rounds batched sizes to multiples of 16 (base_fst.py:693-694)
            self.height, self.width = height // 16 * 16, width // 16 * 16
        self.seed, self.suffix = seed, suffix
        self.sampling_origin_ds, self.sampling_origin_ds_subset = 'SYNTH', 'val'
        self.sampling_cats, self.sampling_scenario, self.finetune = 'novel', 'parents', 'None'

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return episodes.make_episode(int(idx), self.n_ways, self.k_shots, self.height, self.width,
                                     self.spp_img_size, seed=self.seed)

    def reshuffle(self):
        self.seed += self.length

    def evaluate(self, results=None, results_pkl_dir_fp=None, model_dir=None, total=1):
        """Same return keys as BaseFewShotISEG.evaluate (base_fst.py:1597-1601)."""
        assert (results is not None) ^ (results_pkl_dir_fp is not None)
        out = {}
        for kind, key in (('segm', 'isegm'), ('bbox', 'bbox')):
            ev = FSISEGEval(results=results, results_pkl_dir_fp=results_pkl_dir_fp, n_ways=self.n_ways,
                            iou_type=kind).run()
            out[f'{key}_mAP'], out[f'{key}_mAR'] = ev['mAP'], ev['mAR']
        return out


def write_chunked(results_iter, out_dir, chunk=1000):
    """main.py:290-309: results pickled in chunks of 1000 as ResultsChunked/NN.pkl."""
    os.makedirs(out_dir, exist_ok=True)
    buf, counter = [], 0
    for res in results_iter:
        buf.extend(res)
        if len(buf) >= chunk:
            with open(os.path.join(out_dir, f'{counter:02}.pkl'), 'wb') as fh:
                pickle.dump(buf, fh)
            buf, counter = [], counter + 1
    if buf:
        with open(os.path.join(out_dir, f'{counter:02}.pkl'), 'wb') as fh:
            pickle.dump(buf, fh)


class ClutteredCharsFewShotISEG(Dataset):
    """MNISTISEG / OMNIISEG-shaped episodes (cfg1 / cfg2 of BASELINE.json) over the numpy cluttered-
    character generator: the sample dict of base_fst.py:1248-1266, class-major supports built from one
    instance each with the reference's crop geometry (``support_from_instance``: offset ratio from fill ratio 0.8,
    square reflect-padded crop, resize of the longer side, centre pad), per-episode category ids 0..N-1, images normalised with the
    dataset mean/std (datasets/mnistiseg/ParamsMNISTISEG.json:1-5, datasets/omniiseg/ParamsOMNIISEG.json:1-5).
    Character datasets are batched without aspect-ratio grouping (base_fst.py:611-624)."""
    PARAMS = {'MNISTISEG': dict(mean=(0.9531239867210388, 0.9524800777435303, 0.9531603455543518),
                                std=(0.16827817261219025, 0.16883736848831177, 0.16667258739471436), n_cats=10),
              'OMNIISEG': dict(mean=(0.9628916382789612, 0.9640044569969177, 0.9626953601837158),
                               std=(0.16037128865718842, 0.15775758028030396, 0.15985246002674103), n_cats=26)}

    def __init__(self, dataset='MNISTISEG', n_ways=1, k_shots=1, n_imgs=32, img_size=128, spp_img_size=128,
                 spp_fill_ratio=0.8, batch=1, shuffle=False, seed=1234):
        par = self.PARAMS[dataset]
        self.n_ways, self.k_shots, self.batch, self.shuffle = n_ways, k_shots, batch, shuffle
        self.img_size, self.spp_img_size, self.spp_fill_ratio = img_size, spp_img_size, spp_fill_ratio
        self.sampling_origin_ds, self.sampling_origin_ds_subset = dataset, 'val'
        self.sampling_cats, self.sampling_scenario, self.finetune = 'all', 'parents', 'None'
        self.mean = np.asarray(par['mean'], np.float32)
        self.std = np.asarray(par['std'], np.float32)
        self.seed = seed
        cats = np.arange(par['n_cats'])
        self.images = [cc.make_image(seed + j, img_size, cats) for j in range(n_imgs)]
        # instance table: (image, object) per class, instance id = position in this table
        self.inst = [(j, o) for j, im in enumerate(self.images) for o in range(len(im['cat_ids']))]
        self.inst_cat = np.array([self.images[j]['cat_ids'][o] for j, o in self.inst])
        self.cats = cats
        self.order_initial = np.arange(n_imgs, dtype=np.int32)
        self.order = self.order_initial.copy()

    def __len__(self):
        return len(self.order)

    def reshuffle(self, e=8):
        """base_fst.py:605-624."""
        order = list(self.order_initial)
        if self.shuffle:
            random.Random((2 ** e) % 1000).shuffle(order)
        self.order = np.array(order, dtype=np.int32)

    def _norm(self, img_u8):
        return torch.from_numpy(((img_u8.astype(np.float32) / 255.0 - self.mean) / self.std).transpose(2, 0, 1).copy())

    def __getitem__(self, idx):
        child = int(self.order[int(idx)])
        im = self.images[child]
        rng = np.random.RandomState(self.seed * 31 + child)
        # classes of the episode: the query's own classes first, then random others (novel classes of a
        # query always get a support; base_fst.py samples the same way from the query's categories)
        present = list(dict.fromkeys(im['cat_ids'].tolist()))
        rng.shuffle(present)
        have = set(self.inst_cat.tolist())           # only classes with at least one instance can give a support
        others = [c for c in rng.permutation(self.cats).tolist() if c not in present and c in have]
        real = np.array((present + others)[:self.n_ways], np.int64)
        if len(real) < self.n_ways:
            raise ValueError(f'the image pool holds {len(have)} classes, fewer than n_ways={self.n_ways}')
        mapping = np.full(int(self.cats.max()) + 1, -1, np.int64)
        mapping[real] = np.arange(len(real))
        keep = mapping[im['cat_ids']] >= 0
        spp_imgs, spp_boxes, spp_masks, spp_ids = [], [], [], []
        for c in real:                                   # class-major (base_fst.py:1054-1080)
            pool = np.flatnonzero(self.inst_cat == c)
            foreign = np.array([p for p in pool if self.inst[p][0] != child], np.int64)
            pool = foreign if len(foreign) >= 1 else pool
            pick = rng.choice(pool, self.k_shots, replace=len(pool) < self.k_shots)
            for p in pick:
                j, o = self.inst[p]
                src = self.images[j]
                crop, nb, m = support_from_instance(src['img'], src['bboxes'][o], src['isegmaps'][o],
                                                    self.spp_img_size, self.spp_fill_ratio)
                spp_imgs.append(self._norm(crop)); spp_boxes.append(nb); spp_masks.append(m); spp_ids.append(p)
        return {
            'idx': int(idx),
            'qry_child_idx': child,
            'qry_img': self._norm(im['img']),
            'qry_cat_ids_real': im['cat_ids'][keep].astype(np.int64),
            'qry_cat_ids': mapping[im['cat_ids'][keep]].astype(np.int64),
            'qry_bboxes': im['bboxes'][keep].astype(np.float32),
            'qry_isegmaps': im['isegmaps'][keep].astype(bool),
            'spp_imgs': torch.stack(spp_imgs),
            'spp_bboxes': np.stack(spp_boxes).astype(np.float32),
            'spp_isegmaps': np.stack(spp_masks).astype(bool),
            'cats_ids_to_sample_real': real,
            'cats_ids_to_sample': np.arange(len(real), dtype=np.int64),
            'spp_insts_ids': np.asarray(spp_ids, np.int64),
            'img_shape': np.array([self.img_size, self.img_size, 3], dtype=np.int32),
        }

    evaluate = SyntheticFewShotISEG.evaluate


# ------------------------------------------------------------------------------------------------
# Episodic sampling on a databag (which instances form an episode): the index half of
# BaseFewShotISEG.load_dataset / reshuffle / __getitem__ / get_query / get_support
# ------------------------------------------------------------------------------------------------
class Databag:
    """The lookup tables ``load_dataset`` collects or reads (base_fst.py:328-432): parent queries (one per image, with
    ``cats_dict`` category -> instance ids in insertion order and the ids of their child queries), child queries
    ``[parent, category]`` (one per image and category on it), per-class instance lists, per-instance parent / category
    / box.  Built from the reference's pickle (``from_pickle``: the tuple ``(qrys_parents_, qrys_children,
    cats_insts_list, insts)``; the older files hold SETS as class lists - kept in their iteration order, which is what
    the reference's list comprehensions see) or from flat arrays (``from_arrays``: tests/golden/databag_sampler.npz)."""

    def __init__(self, parents_cats, parents_children, children, class_lists, inst_parent, inst_cat, inst_bbox):
        self.parents_cats = parents_cats            # list of [(cat_id, [inst ids])] per parent, insertion order
        self.parents_children = parents_children    # list of [child ids] per parent
        self.children = children                    # [n_children, 2] (parent, category)
        self.class_lists = class_lists              # list of [inst ids] per category
        self.inst_parent, self.inst_cat, self.inst_bbox = inst_parent, inst_cat, inst_bbox

    @classmethod
    def from_pickle(cls, path: str) -> 'Databag':
        with open(path, 'rb') as fh:
            parents, children, class_lists, insts = pickle.load(fh)
        return cls([[(int(c), [int(i) for i in v]) for c, v in p['cats_dict'].items()] for p in parents],
                   [[int(c) for c in p['nums_children_qrys']] for p in parents],
                   np.asarray(children, np.int64).reshape(-1, 2), [[int(i) for i in lst] for lst in class_lists],
                   np.array([int(i.get('num_parent_qry', -1)) for i in insts]),
                   np.array([int(i['cat_id']) for i in insts]),
                   np.array([np.asarray(i['bbox'], np.float32) for i in insts], np.float32).reshape(-1, 4))

    @classmethod
    def from_arrays(cls, z, prefix: str = 'bag__') -> 'Databag':
        g = lambda k: np.asarray(z[prefix + k])
        pp, pc, pi = g('parent_ptr'), g('parent_cat'), g('parent_inst')
        parents_cats = []
        for a, b in zip(pp[:-1], pp[1:]):
            d = {}
            for c, i in zip(pc[a:b], pi[a:b]):
                d.setdefault(int(c), []).append(int(i))
            parents_cats.append(list(d.items()))
        cp, ch = g('parent_children_ptr'), g('parent_children')
        lp, li = g('class_ptr'), g('class_inst')
        return cls(parents_cats, [[int(c) for c in ch[a:b]] for a, b in zip(cp[:-1], cp[1:])], g('children'),
                   [[int(i) for i in li[a:b]] for a, b in zip(lp[:-1], lp[1:])], g('inst_parent'), g('inst_cat'),
                   g('inst_bbox'))


class DatabagEpisodeSampler:
    """Index producer of the reference's episodic dataset on a ``Databag``: ``len`` / ``order`` as ``load_dataset``
    builds them, ``sample_indices(idx)`` = everything ``__getitem__`` decides before it touches a pixel.  Draws come
    from Python's ``random`` module in the reference's call order (``random.choice`` of the child under 'parents',
    ``random.sample`` of the other categories, ``random.shuffle`` of the N, ``random.sample`` of K instances per
    category), so a seeded run replays the reference's episodes exactly (tests/golden/databag_sampler.npz)."""

    def __init__(self, bag: Databag, n_ways: int, k_shots: int, cats_novel, cats_total_amount: int,
                 sampling_cats: str = 'base_', sampling_scenario: str = 'parents', shuffle: bool = False, repeats: int = 1,
                 first_parents__only: int = 0, first_children_only: int = 0, qry_cats_choice_remove: bool = False,
                 qry_cats_choice_random: bool = False, qry_cats_order_shuffle: bool = True,
                 delete_qry_insts_in_spp_insts_on_train: bool = True, spp_random: bool = True, rng=None):
        self.bag, self.n_ways, self.k_shots = bag, int(n_ways), int(k_shots)
        self.sampling_scenario = sampling_scenario
        self.shuffle = shuffle
        self.qry_cats_choice_remove, self.qry_cats_choice_random = qry_cats_choice_remove, qry_cats_choice_random
        self.qry_cats_order_shuffle = qry_cats_order_shuffle
        self.delete_qry_insts = delete_qry_insts_in_spp_insts_on_train
        self.spp_random = spp_random
        self.random = rng if rng is not None else random          # the module itself: the reference's global stream
        # ---- cats_selection (base_fst.py:267-300)
        novel = np.array([] if sampling_cats == 'all' else cats_novel, dtype=np.int32)
        is_base = np.ones(cats_total_amount, bool)
        is_base[novel] = False
        base = np.where(is_base)[0]
        if sampling_cats == 'base_':
            self.cats_to_save = base
        elif sampling_cats == 'novel':
            self.cats_to_save = novel
        elif sampling_cats == 'all':
            self.cats_to_save = np.arange(cats_total_amount).astype(np.int32)
        else:
            raise ValueError(f'sampling_cats {sampling_cats!r}')
        self.cats_to_save_bool = np.zeros(cats_total_amount, bool)
        self.cats_to_save_bool[self.cats_to_save] = True
        # ---- order (base_fst.py:438-474)
        n_par, n_chi = len(bag.parents_cats), len(bag.children)
        if sampling_scenario not in ('parents', 'children'):
            raise ValueError(f'sampling_scenario {sampling_scenario!r}')
        order = np.arange(n_par if sampling_scenario == 'parents' else n_chi)
        if 0 < first_parents__only <= n_par:
            if sampling_scenario == 'parents':
                order = order[:first_parents__only]
            else:
                order = order[:bag.parents_children[first_parents__only - 1][-1] + 1]
        if 0 < first_children_only <= n_chi:
            if sampling_scenario == 'parents':
                order = order[:int(bag.children[first_children_only - 1][0]) + 1]
            else:
                order = order[:first_children_only]
        if not 1 <= repeats <= 100:
            repeats = 1
        self.order_initial = np.tile(order, reps=repeats)
        self.reshuffle()

    def reshuffle(self, e: int = 8) -> None:
        """``reshuffle`` of a batch-1 / synthetic-dataset run (base_fst.py:611-625): a fixed permutation per epoch seed."""
        self.order = self.order_initial.copy()
        if self.shuffle:
            order = list(self.order)
            random.Random((2 ** e) % 1000).shuffle(order)
            self.order = np.array(order, dtype=np.int32)

    def __len__(self) -> int:
        return len(self.order)

    def sample_indices(self, idx: int) -> dict:
        bag, R = self.bag, self.random
        real_idx = int(self.order[idx])
        # __getitem__ (base_fst.py:1197-1210)
        child = R.choice(bag.parents_children[real_idx]) if self.sampling_scenario == 'parents' else real_idx
        parent, cat_main = int(bag.children[child][0]), int(bag.children[child][1])
        cats_dict = dict(bag.parents_cats[parent])
        cats_on_img = list(cats_dict)
        # get_query (base_fst.py:793-824): the N - 1 other categories
        cats = [cat_main]
        keep = self.cats_to_save_bool.copy()
        keep[cat_main] = False
        if self.qry_cats_choice_remove:
            keep[cats_on_img] = False
        pool = [int(c) for c in np.nonzero(keep)[0]]
        if self.qry_cats_choice_random:
            if len(pool) < self.n_ways - 1:
                raise NotImplementedError(f'could not select {self.n_ways} categories')
            other = R.sample(pool, self.n_ways - 1)
        else:
            other = pool[:self.n_ways - 1]
        cats.extend(other)
        if self.qry_cats_order_shuffle:
            R.shuffle(cats)
        cats_real = np.array(cats, dtype=np.int32)
        # ... the query's instances (base_fst.py:826-846)
        qry_insts, qry_cats = [], []
        for c in cats_real:
            if int(c) not in cats_dict:
                continue
            qry_insts.extend(cats_dict[int(c)])
            qry_cats.extend([int(c)] * len(cats_dict[int(c)]))
        # get_support (base_fst.py:1052-1080): K instances per category
        spp = []
        for c in cats_real:
            lst = bag.class_lists[int(c)]
            pool_i = [v for v in lst if v not in qry_insts] if self.delete_qry_insts else lst
            if self.spp_random:
                if len(pool_i) < self.k_shots:
                    raise NotImplementedError(f'could not sample {self.k_shots} instances of category {int(c)}')
                spp.extend(R.sample(pool_i, self.k_shots))
            else:
                spp.extend(pool_i[:self.k_shots])
        # remap to 0..N-1 (base_fst.py:1243-1246)
        mapping = np.zeros(int(cats_real.max()) + 1, dtype=np.int32)
        mapping[cats_real] = np.arange(len(cats_real))
        qry_cats = np.array(qry_cats, dtype=np.int32)
        qry_insts = np.array(qry_insts, dtype=np.int32)
        return dict(idx=idx, qry_child_idx=int(child), idx_parent=parent, cat_id_main=cat_main,
                    cats_ids_to_sample_real=cats_real.astype(np.int64),
                    cats_ids_to_sample=mapping[cats_real].astype(np.int64),
                    qry_insts_ids=qry_insts, qry_cat_ids_real=qry_cats.astype(np.int64),
                    qry_cat_ids=mapping[qry_cats].astype(np.int64) if len(qry_cats) else np.zeros(0, np.int64),
                    qry_bboxes=bag.inst_bbox[qry_insts].astype(np.float32).reshape(-1, 4),
                    spp_insts_ids=np.array(spp).astype(np.int64))
