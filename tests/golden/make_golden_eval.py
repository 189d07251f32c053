#!/usr/bin/env python3
"""Golden vectors for the evaluator's own code: datasets/fewshotiseg/fsisegeval.py (``FSISEGEval.__init__``:
reading the chunked result pickles, YXYX -> [x, y, max(w,1), max(h,1)] boxes, annotation ids, COCOeval parameters;
``_prepare``: grouping by (image, category); ``summarize_short``: means over the entries > -1).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_eval.py

fsisegeval.py is imported unmodified.  pycocotools is not installed: ``COCOeval`` is a bare base class holding
``params`` / ``_gts`` / ``_dts`` (everything ``evaluate`` / ``accumulate`` do is pycocotools and stays UNPINNED),
``pycocotools.mask.area`` is the pixel count of the decoded RLE.  The result dicts are seeded, written as two chunk
files the way main.py:290-309 writes them, and read back by the reference through its own ``read_pkl``.
Nothing from the reference is copied: this script imports it, feeds seeded inputs and stores outputs."""
import os
import pickle
import sys
import tempfile
import types
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)


def make_results(seed=20261005):
    """Five seeded ``simple_test`` result dicts (keys fgn.py:276-302) over 24x40 images, 3 ways."""
    from fgn_amd import rle
    rng = np.random.RandomState(seed)
    H, W = 24, 40
    out = []
    for i in range(5):
        ng, nd = int(rng.randint(0, 4)), int(rng.randint(0, 6))
        box = lambda n: np.stack([rng.rand(n) * 10, rng.rand(n) * 15, 12 + rng.rand(n) * 12, 20 + rng.rand(n) * 20], 1).astype(np.float32)
        out.append({'idx': np.int64(i), 'qry_img_shape': np.array([H, W, 3], np.int32),
                    'qry_bboxes': box(ng), 'qry_cat_ids': rng.randint(0, 3, ng).astype(np.int64),
                    'qry_isegmaps_rle': [rle.encode(rng.rand(H, W) > 0.6) for _ in range(ng)],
                    'dt_bboxes': box(nd), 'dt_cat_ids': rng.randint(0, 3, nd).astype(np.int64),
                    'dt_scores': rng.rand(nd).astype(np.float32),
                    'dt_isegmaps_rle': [rle.encode(rng.rand(H, W) > 0.5) for _ in range(nd)]})
    out[3]['qry_bboxes'][0] = [3.0, 4.0, 3.5, 4.25]          # thinner than one pixel: w, h clamp to 1
    return out


def eval_arrays(seed=5):
    """A synthetic COCOeval.eval: precision [T=1, R=11, K=3, A=1, M=1] and recall [T, K, A, M] with -1 entries."""
    rng = np.random.RandomState(seed)
    p = rng.rand(1, 11, 3, 1, 1)
    p[0, :, 1] = -1.0
    r = rng.rand(1, 3, 1, 1)
    r[0, 1] = -1.0
    return p, r


def main():
    import make_golden_data as G
    from fgn_amd import rle
    for name in ('pycocotools', 'pycocotools.mask', 'pycocotools.cocoeval', 'mmdet', 'mmdet.core'):
        sys.modules[name] = G._AnyModule(name)

    class COCOeval:                                   # bare stand-in for the pycocotools base class
        def __init__(self, cocoGt=None, cocoDt=None, iouType='segm'):
            self.params = types.SimpleNamespace(iouType=iouType)
            self._gts, self._dts = defaultdict(list), defaultdict(list)
    sys.modules['pycocotools.cocoeval'].COCOeval = COCOeval
    sys.modules['pycocotools.mask'].area = lambda r: int(rle.decode(r).sum())
    for k in [k for k in sys.modules if k == 'datasets' or k.startswith('datasets.')]:
        del sys.modules[k]
    for name, path in (('datasets', 'datasets'), ('datasets.fewshotiseg', 'datasets/fewshotiseg')):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m
    sys.path.insert(0, REF)
    from datasets.fewshotiseg import fsisegeval as ref

    store = {}
    results = make_results()
    with tempfile.TemporaryDirectory() as d:
        for name, chunk in (('00.pkl', results[:3]), ('01.pkl', results[3:])):
            with open(os.path.join(d, name), 'wb') as fh:
                pickle.dump(chunk, fh)
        for kind in ('segm', 'bbox'):
            ev = ref.FSISEGEval(results_pkl_dir_fp=d, n_ways=3, iou_type=kind)
            ev._prepare()
            pre = kind + '__'
            store[pre + 'imgs'] = np.array([[im['height'], im['width']] for im in ev.imgs])
            for tag, anns in (('gt', ev.gts), ('dt', ev.dts)):
                store[pre + tag + '_image_id'] = np.array([a['image_id'] for a in anns])
                store[pre + tag + '_id'] = np.array([a['id'] for a in anns])
                store[pre + tag + '_bbox'] = np.array([a['bbox'] for a in anns], np.float64).reshape(-1, 4)
                store[pre + tag + '_category_id'] = np.array([a['category_id'] for a in anns])
                store[pre + tag + '_area'] = np.array([a['area'] for a in anns])
            store[pre + 'dt_score'] = np.array([a['score'] for a in ev.dts])
            store[pre + 'gt_flags'] = np.array([[int(a['iscrowd']), int(bool(a['ignore']))] for a in ev.gts]).reshape(-1, 2)
            pr = ev.params
            store[pre + 'recThrs'] = np.asarray(pr.recThrs)
            store[pre + 'scalars'] = np.array([pr.iouThrs[0], pr.maxDets[0], pr.areaRng[0][0], pr.areaRng[0][1], pr.useCats])
            store[pre + 'imgIds'], store[pre + 'catIds'] = np.asarray(pr.imgIds), np.asarray(pr.catIds)
            keys = sorted(ev._gts) + sorted(ev._dts)
            store[pre + 'group_keys'] = np.array(keys).reshape(-1, 2)
            store[pre + 'group_sizes'] = np.array([len(ev._gts[k]) for k in sorted(ev._gts)] + [len(ev._dts[k]) for k in sorted(ev._dts)])
            store[pre + 'n_gt_groups'] = np.array(len(ev._gts))
    p, r = eval_arrays()
    ev.eval = {'precision': p, 'recall': r}
    out = ev.summarize_short()
    store['summary'] = np.array([out['mAP'], out['mAR']])
    ev.eval = {'precision': -np.ones_like(p), 'recall': -np.ones_like(r)}
    out = ev.summarize_short()
    store['summary_empty'] = np.array([out['mAP'], out['mAR']], np.float64)
    np.savez_compressed(os.path.join(HERE, 'fsiseg_eval.npz'), **store)
    print('written', os.path.join(HERE, 'fsiseg_eval.npz'), len(store), 'arrays')


if __name__ == '__main__':
    main()
