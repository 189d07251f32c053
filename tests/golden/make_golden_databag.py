#!/usr/bin/env python3
"""Golden vectors for the EPISODIC SAMPLER of the reference (which instances form an episode) on its own shipped databag:

  datasets/fewshotiseg/base_fst.py:267-300     cats_selection (base / novel split of the 26 OMNIISEG letters; the novel
                                               set is the one the shipped file was written under, see NOVEL below)
  datasets/fewshotiseg/base_fst.py:300-486     load_dataset on an EXISTING databag pickle
                                               (resources/omniiseg_fst/OMNIISEG2OMNIISEG_OMNIISEG_val_base_.pkl: 968 parents,
                                               1736 children, 26 class lists, 1792 instances): order by parents /
                                               children, first_parents__only / first_children_only cuts, repeats
  datasets/fewshotiseg/base_fst.py:605-625     reshuffle, batch == 1 branch (random.Random((2 ** e) % 1000).shuffle)
  datasets/fewshotiseg/base_fst.py:1172-1246   __getitem__: index -> child query (random.choice among the parent's
                                               children under 'parents') -> parent image + main category
  datasets/fewshotiseg/base_fst.py:793-824     get_query: the N - 1 other categories (first ones or random.sample,
                                               optionally without the categories on the image), random.shuffle of the N
  datasets/fewshotiseg/base_fst.py:826-846     ... the query's instance ids / category ids / boxes
  datasets/fewshotiseg/base_fst.py:1052-1080   get_support: the K support instance ids per category (random.sample of the
                                               class list minus the query's own instances, or the first K)
  datasets/fewshotiseg/base_fst.py:1243-1246   remap of the real category ids to 0..N-1

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_databag.py

The reference's files are imported unmodified (same stand-in modules as make_golden_data.py).  The dataset object is made
with ``__new__`` and given the attributes ``__init__`` + ``select_cats`` would set; ``load_dataset`` / ``reshuffle`` /
``__getitem__`` / ``get_query`` / ``get_support`` then run as they are.  What they would read from image files is
answered by stand-ins that the INDEX logic does not depend on: ``cv2.imread`` gives a blank 256 x 256 image (OMNIISEG's
size; ``get_new_shape(256, 256, 256, 256)`` keeps it, so no resize), ``get_isegmap`` a blank mask, the two imgaug
operators of get_support blank crops.  ``random`` is seeded per item so that every draw can be replayed.
Stored: the databag's index structure (no image data: children table, per-parent category -> instance lists, per-class
instance lists, per-instance parent / category / box) and, per configuration and item, everything the sampler decides.
Nothing from the reference is copied: this script imports it, feeds its own data file and stores outputs.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_data as mgd  # noqa: E402  (stand-in modules + import of the reference)

PKL = os.path.join(mgd.REF, 'datasets/fewshotiseg/resources/omniiseg_fst/OMNIISEG2OMNIISEG_OMNIISEG_val_base_.pkl')

# configurations of the sampler (attribute name -> value); `items` dataset indices are drawn per configuration
CONFIGS = {
    # the evaluation-style set-up of fgn_train.py's eval_ds_cfg: children in order, fixed N - 1 other categories
    'eval_children': dict(n_ways=3, k_shots=1, sampling_scenario='children', shuffle=False, repeats=1,
                          first_parents__only=0, first_children_only=0, qry_cats_choice_random=False,
                          qry_cats_choice_remove=False, qry_cats_order_shuffle=True, spp_random=True,
                          delete_qry_insts_in_spp_insts_on_train=True),
    # the training-style set-up of fgn_train.py / fgn_ft.py: parents, shuffled order, random other categories
    'train_parents': dict(n_ways=3, k_shots=3, sampling_scenario='parents', shuffle=True, repeats=2,
                          first_parents__only=400, first_children_only=0, qry_cats_choice_random=True,
                          qry_cats_choice_remove=False, qry_cats_order_shuffle=True, spp_random=True,
                          delete_qry_insts_in_spp_insts_on_train=True),
    'one_way_first_k': dict(n_ways=1, k_shots=2, sampling_scenario='parents', shuffle=False, repeats=1,
                            first_parents__only=0, first_children_only=300, qry_cats_choice_random=False,
                            qry_cats_choice_remove=False, qry_cats_order_shuffle=False, spp_random=False,
                            # (the class lists of this file are SETS, the older databag format: `insts_pool[:k]` needs the
                            # list the delete-branch builds; the golden stores every class list in the set's iteration order)
                            delete_qry_insts_in_spp_insts_on_train=True),
    'five_way_remove': dict(n_ways=5, k_shots=2, sampling_scenario='children', shuffle=True, repeats=1,
                            first_parents__only=0, first_children_only=0, qry_cats_choice_random=True,
                            qry_cats_choice_remove=True, qry_cats_order_shuffle=True, spp_random=True,
                            delete_qry_insts_in_spp_insts_on_train=True),
}
N_ITEMS = 48
# The shipped pickle predates omniiseg_fst.py's current split (novel = the letters of 'SPUTNIK'): the classes WITHOUT an
# instance list in this 'base_' bag are B, D, H, I, N, V.  With the current split `cats_to_save` would contain three empty
# classes and get_support raises NotImplementedError on the first of them; the sampler is therefore exercised with the
# split the file was written under (select_cats only assigns `cats_novel`; everything downstream is cats_selection's).
NOVEL = [1, 3, 7, 8, 13, 21]
IMG = 256          # OMNIISEG image size (datasets/omniiseg: 256 x 256 renders)


def item_indices(length: int, tag: str) -> np.ndarray:
    rng = np.random.RandomState(abs(hash(tag)) % (2 ** 31) if False else sum(map(ord, tag)))
    return np.sort(rng.choice(length, size=min(N_ITEMS, length), replace=False))


def item_seed(tag: str, idx: int) -> int:
    return sum(map(ord, tag)) * 100003 + int(idx)


class _Box:
    y1 = x1 = 0.0
    y2 = x2 = 1.0


def make_dataset(base_fst, cfg: dict):
    B = base_fst.BaseFewShotISEG
    ds = B.__new__(B)
    # what BaseFewShotISEG.__init__ (base_fst.py:172-265) and OMNIFewShotISEG.select_cats (omniiseg_fst.py:14-31) set
    ds.verbose = False
    for k, v in cfg.items():
        setattr(ds, k, v)
    ds.cats_total_amount = 26
    ds.cats_novel = np.array(NOVEL, dtype=np.int32)
    ds.sampling_cats = 'base_'
    ds.finetune = 'Ignore'
    ds.sampling_origin_ds = 'OMNIISEG'
    ds.batch = 1
    ds.merged_ds = ds.upper_ds = None
    ds.databag_fp = PKL
    ds.spp_img_size, ds.target_size, ds.max_size = 128, IMG, IMG
    ds.spp_fill_ratio, ds.spp_crop_square = 0.8, True
    ds.offset_ratio = np.around(1 / (2 * ds.spp_fill_ratio) - 0.5, decimals=2)
    ds.augment_qry = ds.augment_spp = False
    ds.transforms = None
    ds.get_plot = ds.overfit_sample_mode = False
    ds.overfit_sample = None
    ds.imgs_dir_fp = '/nonexistent'
    ds.imgs_dir_fps = None
    ds.a_print = ds.v_print = ds.e_print = lambda *a, **k: None
    ds.load_dataset()
    # stand-ins for what reads pixels (the index logic does not look at any of it)
    blank = np.zeros((IMG, IMG, 3), np.uint8)
    base_fst.cv2.imread = lambda *a, **k: blank
    ds.get_isegmap = lambda img, bbox, info: np.zeros(img.shape[:2], bool)
    ds.get_bboxes_on_img_from_yxyx = lambda *a, **k: None
    S = ds.spp_img_size
    ds.resize_spp = lambda image, bounding_boxes: (np.zeros((S, S) + image.shape[2:], image.dtype), [_Box()])
    ds.pad_spp = lambda image, bounding_boxes: (np.zeros((S, S) + image.shape[2:], image.dtype), [_Box()])
    return ds


def databag_index(ds) -> dict:
    """The index structure of the databag as flat arrays (no image data)."""
    st = {}
    st['children'] = np.array(ds.qrys_children, dtype=np.int64).reshape(-1, 2)
    # parents: ragged list of (cat_id, inst_id) in dict insertion order + the parent's children
    pc_ptr, pc_cat, pc_inst, ch_ptr, ch = [0], [], [], [0], []
    for p in ds.qrys_parents_:
        for cat, insts in p['cats_dict'].items():
            for i in insts:
                pc_cat.append(int(cat)); pc_inst.append(int(i))
        pc_ptr.append(len(pc_cat))
        ch.extend(int(c) for c in p['nums_children_qrys'])
        ch_ptr.append(len(ch))
    st.update(parent_ptr=np.array(pc_ptr), parent_cat=np.array(pc_cat), parent_inst=np.array(pc_inst),
              parent_children_ptr=np.array(ch_ptr), parent_children=np.array(ch))
    cl_ptr, cl = [0], []
    for lst in ds.cats_insts_list:
        cl.extend(int(i) for i in lst)
        cl_ptr.append(len(cl))
    st.update(class_ptr=np.array(cl_ptr), class_inst=np.array(cl))
    st['inst_parent'] = np.array([int(i.get('num_parent_qry', -1)) for i in ds.insts])
    st['inst_cat'] = np.array([int(i['cat_id']) for i in ds.insts])
    st['inst_bbox'] = np.array([np.asarray(i['bbox'], np.float32) for i in ds.insts], np.float32).reshape(-1, 4)
    return {'bag__' + k: v for k, v in st.items()}


def main():
    _, base_fst = mgd.import_reference()
    store = {}
    bag_done = False
    for tag, cfg in CONFIGS.items():
        ds = make_dataset(base_fst, cfg)
        if not bag_done:
            store.update(databag_index(ds))
            store['cats_novel'] = np.asarray(ds.cats_novel)
            store['cats_to_save'] = np.asarray(ds.cats_to_save)
            bag_done = True
        store[f'{tag}__order'] = np.asarray(ds.order, dtype=np.int64)
        store[f'{tag}__len'] = np.array(len(ds))
        items = item_indices(len(ds), tag)
        store[f'{tag}__items'] = items
        rows = {k: [] for k in ('qry_child_idx', 'cats_ids_to_sample_real', 'cats_ids_to_sample', 'spp_insts_ids')}
        ragged = {k: [] for k in ('qry_cat_ids_real', 'qry_cat_ids', 'qry_bboxes')}
        ptr = [0]
        for idx in items:
            random.seed(item_seed(tag, idx))
            s = ds[int(idx)]
            assert s['spp_imgs'].shape[0] == cfg['n_ways'] * cfg['k_shots']
            for k in rows:
                rows[k].append(np.asarray(s[k]))
            for k in ragged:
                ragged[k].append(np.asarray(s[k]))
            ptr.append(ptr[-1] + len(s['qry_cat_ids']))
        for k, v in rows.items():
            store[f'{tag}__{k}'] = np.stack(v)
        for k, v in ragged.items():
            store[f'{tag}__{k}'] = np.concatenate(v)
        store[f'{tag}__qry_ptr'] = np.array(ptr)
    np.savez_compressed(os.path.join(HERE, 'databag_sampler.npz'), **store)
    # self-check (build container only): the product's pickle reader sees the same tables as the stored arrays
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from fgn_amd.fewshot_ds import Databag
    a, b = Databag.from_pickle(PKL), Databag.from_arrays(np.load(os.path.join(HERE, 'databag_sampler.npz')))
    assert a.parents_cats == b.parents_cats and a.parents_children == b.parents_children and a.class_lists == b.class_lists
    assert np.array_equal(a.children, b.children) and np.array_equal(a.inst_bbox, b.inst_bbox)
    assert np.array_equal(a.inst_parent, b.inst_parent) and np.array_equal(a.inst_cat, b.inst_cat)
    print('written', os.path.join(HERE, 'databag_sampler.npz'), len(store), 'arrays')


if __name__ == '__main__':
    main()
