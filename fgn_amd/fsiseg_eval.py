"""FSISEGEval: few-shot instance-segmentation AP/AR at IoU 0.5.

Restatement of the reference's evaluator (datasets/fewshotiseg/fsisegeval.py:14-185), which
is pycocotools ``COCOeval`` with: one IoU threshold 0.5, 11 recall thresholds 0:0.1:1,
maxDets 100, one area range, per-episode category ids 0..N-1, no crowd/ignore
(fsisegeval.py:108-116).  Consumes the result dicts ``FGN.simple_test`` returns (keys
fgn.py:276-302), directly or from the chunked pickles the eval hook writes
(main.py:290-309).  numpy only; pycocotools is not available in the build image, so the
evaluate/accumulate semantics are restated from its published behaviour (parity unpinned).
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, Iterable, List

import numpy as np

from . import rle as _rle


def _rle_area(r: dict) -> int:
    return int(_rle.decode(r).sum())


def _mask_iou(dts: List[dict], gts: List[dict]) -> np.ndarray:
    """pycocotools maskUtils.iou for non-crowd RLEs: |d & g| / |d | g|.  Intersections are one
    float32 matrix product over the pixels of the joint bounding box of all masks (exact: counts
    stay far below 2^24), in column chunks to bound memory."""
    if not dts or not gts:
        return np.zeros((len(dts), len(gts)))
    d = np.stack([_rle.decode(x).astype(bool) for x in dts])
    g = np.stack([_rle.decode(x).astype(bool) for x in gts])
    occ = d.any(0) | g.any(0)
    if not occ.any():
        return np.zeros((len(dts), len(gts)))
    ys, xs = np.flatnonzero(occ.any(1)), np.flatnonzero(occ.any(0))
    d = d[:, ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1].reshape(len(dts), -1)
    g = g[:, ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1].reshape(len(gts), -1)
    inter = np.zeros((len(dts), len(gts)), np.float64)
    for c0 in range(0, d.shape[1], 1 << 18):
        inter += d[:, c0:c0 + (1 << 18)].astype(np.float32) @ g[:, c0:c0 + (1 << 18)].astype(np.float32).T
    union = d.sum(-1, dtype=np.float64)[:, None] + g.sum(-1, dtype=np.float64)[None, :] - inter
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.where(union > 0, inter / union, 0.0)


def _bbox_iou(d: np.ndarray, g: np.ndarray) -> np.ndarray:
    """pycocotools bbIou on [x, y, w, h], non-crowd."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    dx2, dy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3]
    gx2, gy2 = g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    w = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    h = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = w * h
    union = (d[:, 2] * d[:, 3])[:, None] + (g[:, 2] * g[:, 3])[None] - inter
    return inter / union


def _xywh(yxyx: np.ndarray) -> np.ndarray:
    """fsisegeval.py:63-69 / 85-90: YXYX -> [x, y, max(w,1), max(h,1)]."""
    b = np.asarray(yxyx, np.float64).reshape(-1, 4)
    return np.column_stack((b[:, 1], b[:, 0], np.maximum(b[:, 3] - b[:, 1], 1), np.maximum(b[:, 2] - b[:, 0], 1)))


class FSISEGEval:
    def __init__(self, results: Iterable[Dict] = None, results_pkl_dir_fp: str = None, n_ways: int = 3,
                 iou_type: str = 'segm', iou_thr: float = 0.5):
        if results is None:
            results = []
            for f in sorted(os.listdir(results_pkl_dir_fp)):
                with open(os.path.join(results_pkl_dir_fp, f), 'rb') as fh:
                    results.extend(pickle.load(fh))
        self.results = list(results)
        self.n_ways = n_ways
        self.iou_type = iou_type
        self.iou_thr = iou_thr          # the reference evaluates at 0.5 only (fsisegeval.py:108-116)
        self.rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .10)) + 1, endpoint=True)
        self.max_dets = 100
        self.eval = {}

    def _evaluate_img(self, res: dict, cat: int):
        """COCOeval.evaluateImg for one (image, category): greedy matching of score-sorted
        detections to the unmatched ground truth of highest IoU >= thr."""
        g_sel = np.flatnonzero(np.asarray(res['qry_cat_ids']).reshape(-1) == cat)
        d_sel = np.flatnonzero(np.asarray(res['dt_cat_ids']).reshape(-1) == cat)
        if len(g_sel) == 0 and len(d_sel) == 0:
            return None
        scores = np.asarray(res['dt_scores'], np.float64).reshape(-1)[d_sel]
        order = np.argsort(-scores, kind='mergesort')[:self.max_dets]
        d_sel, scores = d_sel[order], scores[order]
        if self.iou_type == 'segm':
            ious = _mask_iou([res['dt_isegmaps_rle'][i] for i in d_sel], [res['qry_isegmaps_rle'][i] for i in g_sel])
        else:
            ious = _bbox_iou(_xywh(res['dt_bboxes'])[d_sel], _xywh(res['qry_bboxes'])[g_sel])
        thr = min(self.iou_thr, 1 - 1e-10)
        gtm = np.zeros(len(g_sel), bool)
        dtm = np.zeros(len(d_sel), bool)
        for di in range(len(d_sel)):
            best, m = thr, -1
            for gi in range(len(g_sel)):
                if gtm[gi] or ious[di, gi] < best:
                    continue
                best, m = ious[di, gi], gi
            if m >= 0:
                gtm[m] = True
                dtm[di] = True
        return scores, dtm, len(g_sel)

    def evaluate(self):
        self._per = {(i, k): self._evaluate_img(r, k) for i, r in enumerate(self.results) for k in range(self.n_ways)}

    def accumulate(self):
        R = len(self.rec_thrs)
        precision = -np.ones((R, self.n_ways))
        recall = -np.ones(self.n_ways)
        for k in range(self.n_ways):
            E = [self._per[(i, k)] for i in range(len(self.results)) if self._per[(i, k)] is not None]
            if not E:
                continue
            scores = np.concatenate([e[0] for e in E])
            inds = np.argsort(-scores, kind='mergesort')
            dtm = np.concatenate([e[1] for e in E])[inds]
            npig = sum(e[2] for e in E)
            if npig == 0:
                continue
            tp = np.cumsum(dtm).astype(np.float64)
            fp = np.cumsum(~dtm).astype(np.float64)
            nd = len(tp)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            recall[k] = rc[-1] if nd else 0
            pr = pr.tolist()
            for i in range(nd - 1, 0, -1):
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            q = np.zeros(R)
            pos = np.searchsorted(rc, self.rec_thrs, side='left')
            for ri, pi in enumerate(pos):
                if pi < nd:
                    q[ri] = pr[pi]
            precision[:, k] = q
        self.eval = {'precision': precision, 'recall': recall}

    def summarize_short(self) -> Dict[str, float]:
        """fsisegeval.py:151-185."""
        s = self.eval['precision']
        m_ap = float(np.mean(s[s > -1])) if (s > -1).any() else 0.0
        r = self.eval['recall']
        m_ar = float(np.mean(r[r > -1])) if (r > -1).any() else 0.0
        self.stats = [m_ap, m_ar]
        return {'mAP': m_ap, 'mAR': m_ar}

    def run(self) -> Dict[str, float]:
        self.evaluate()
        self.accumulate()
        return self.summarize_short()


def evaluate_results(results: List[Dict], n_ways: int, iou_thr: float = 0.5) -> Dict[str, float]:
    """bbox and segm mAP/mAR of a list of ``simple_test`` result dicts at one IoU threshold (keys are named
    ``*_mAP50`` at the reference's threshold 0.5, ``*_mAP<thr*100>`` otherwise)."""
    out = {}
    tag = f'mAP{int(round(iou_thr * 100))}'
    for kind in ('bbox', 'segm'):
        r = FSISEGEval(results=results, n_ways=n_ways, iou_type=kind, iou_thr=iou_thr).run()
        out[f'{kind}_{tag}'] = r['mAP']
        out[f'{kind}_mAR'] = r['mAR']
    return out


def as_ground_truth(reference: List[Dict], scored: List[Dict]) -> List[Dict]:
    """Result dicts of ``scored`` with the detections of ``reference`` installed as ground truth: AP of one
    implementation's detections against another's (1.0 = every detection reproduced at the IoU threshold)."""
    out = []
    for c, h in zip(reference, scored):
        r = dict(h)
        r['qry_bboxes'], r['qry_cat_ids'], r['qry_isegmaps_rle'] = c['dt_bboxes'], c['dt_cat_ids'], c['dt_isegmaps_rle']
        out.append(r)
    return out
