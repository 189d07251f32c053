"""Matching of HIP detections to the oracle's on one episode, and the maxima north_star's tolerance is
stated on (BASELINE.json: boxes/labels bit-exact given the same head outputs, masks/scores within 1e-4 fp32).

End to end the two paths accumulate in different orders (MFMA tiles vs MKL), so a selection step
(top-6000, NMS keep, score > 0.05, top-100) can flip for an element whose key sits within ~1e-6 of a
threshold or of a neighbour.  A detection pair is *matched* when the boxes agree within 1e-2 px, with the
same label; every oracle detection without such a partner is reported as a *selection flip* (counted,
never hidden in a percentage).  Host-side numpy; used by the parity tests and by bench.py's accuracy leg."""
from __future__ import annotations

import numpy as np


def match_detections(ref_boxes, ref_labels, got_boxes, got_labels, box_tol: float = 1e-2):
    """One-to-one matching, both box lists [n,4] in the same coordinate order.  Returns (pairs [(i_ref, j_got)],
    unmatched_ref, unmatched_got)."""
    ref_boxes, got_boxes = np.asarray(ref_boxes, np.float64), np.asarray(got_boxes, np.float64)
    pairs, used = [], set()
    for i in range(len(ref_boxes)):
        if len(got_boxes) == 0:
            break
        d = np.abs(got_boxes - ref_boxes[i]).max(1)
        d[np.asarray(got_labels) != ref_labels[i]] = np.inf
        for j in used:
            d[j] = np.inf
        j = int(np.argmin(d))
        if d[j] <= box_tol:
            pairs.append((i, j))
            used.add(j)
    mr = sorted(set(range(len(ref_boxes))) - {i for i, _ in pairs})
    mg = sorted(set(range(len(got_boxes))) - used)
    return pairs, mr, mg


def episode_maxima(ref: dict, got: dict, ref_prob, got_prob, ref_logits=None, got_logits=None) -> dict:
    """ref/got: result dicts of one image (dt_bboxes YXYX, dt_scores, dt_cat_ids); *_prob: [D,14,14] mask
    probabilities in detection order (numpy).  Returns the measured maxima on matched pairs + flip counts."""
    pairs, mr, mg = match_detections(ref['dt_bboxes'], ref['dt_cat_ids'], got['dt_bboxes'], got['dt_cat_ids'])
    out = dict(n_ref=len(ref['dt_scores']), n_got=len(got['dt_scores']), matched=len(pairs),
               flips_ref=len(mr), flips_got=len(mg), max_dscore=0.0, max_dprob=0.0, max_dbox=0.0, max_dlogit=0.0)
    if pairs:
        i, j = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        out['max_dscore'] = float(np.abs(ref['dt_scores'][i].astype(np.float64) - got['dt_scores'][j]).max())
        out['max_dbox'] = float(np.abs(ref['dt_bboxes'][i].astype(np.float64) - got['dt_bboxes'][j]).max())
        rp = np.asarray(ref_prob, np.float64).reshape(len(ref['dt_scores']), -1)
        gp = np.asarray(got_prob, np.float64).reshape(-1, rp.shape[1])
        out['max_dprob'] = float(np.abs(rp[i] - gp[j]).max())
        if ref_logits is not None:
            rl = np.asarray(ref_logits, np.float64).reshape(len(ref['dt_scores']), -1)
            gl = np.asarray(got_logits, np.float64).reshape(-1, rl.shape[1])
            out['max_dlogit'] = float(np.abs(rl[i] - gl[j]).max())
            out['max_abs_logit'] = float(np.abs(rl).max())
    return out


def _pair_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """IoU matrix of two XYXY box lists (float64)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-300)


def proposal_set_difference(ref_props, got_props, iou_thr: float, max_per_img: int, box_tol: float = 1e-2,
                            near: float = 1e-4) -> dict:
    """The proposal lists of the two paths ([n,5] rows x1,y1,x2,y2,score, score-sorted) as SETS: one-to-one matching
    by box (within ``box_tol`` px), the symmetric difference, and for every proposal without a partner the reason it
    can legitimately differ between two fp32 accumulation orders:

    * ``nms_threshold``: its IoU with a higher-scoring proposal of the OTHER list is within ``near`` of the NMS
      threshold (suppressed on one side of the threshold, kept on the other);
    * ``cascade``: it overlaps (IoU > threshold) a proposal of the other list that itself has no partner (that one was
      kept there and suppressed this one; the root flip carries its own reason);
    * ``tail``: it is among the last rows of a full list (a flip further up shifted the cut at ``max_per_img``);
    * ``unexplained`` otherwise - a parity defect, asserted to be absent by the tests.
    Returns counts, the maxima on matched pairs and the unmatched rows with their reason."""
    ref = np.asarray(ref_props, np.float64).reshape(-1, 5)
    got = np.asarray(got_props, np.float64).reshape(-1, 5)
    pairs, used = [], set()
    if len(ref) and len(got):
        d = np.abs(ref[:, None, :4] - got[None, :, :4]).max(2)
        for i in np.argsort(d.min(1), kind='stable'):
            order = np.argsort(d[i], kind='stable')
            for j in order[:4]:
                if d[i, j] > box_tol:
                    break
                if int(j) not in used:
                    pairs.append((int(i), int(j)))
                    used.add(int(j))
                    break
    only_ref = sorted(set(range(len(ref))) - {i for i, _ in pairs})
    only_got = sorted(set(range(len(got))) - used)
    out = dict(n_ref=len(ref), n_got=len(got), matched=len(pairs), only_ref=[], only_got=[], max_dbox=0.0,
               max_dscore=0.0, unexplained=0)
    if pairs:
        i, j = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        out['max_dbox'] = float(np.abs(ref[i, :4] - got[j, :4]).max())
        out['max_dscore'] = float(np.abs(ref[i, 4] - got[j, 4]).max())
    n_flips = len(only_ref) + len(only_got)

    def explain(own, other, own_un, other_un, rows):
        if not own_un:
            return
        iou = _pair_iou(own[own_un, :4], other[:, :4]) if len(other) else np.zeros((len(own_un), 0))
        for k, idx in enumerate(own_un):
            higher = other[:, 4] >= own[idx, 4] - 1e-6 if len(other) else np.zeros(0, bool)
            reason, margin = 'unexplained', None
            if higher.any():
                m = np.abs(iou[k][higher] - iou_thr)
                if m.min() <= near:
                    reason, margin = 'nms_threshold', float(m.min())
            if reason == 'unexplained' and other_un and (iou[k][other_un] > iou_thr).any():
                reason = 'cascade'
            if reason == 'unexplained' and len(own) >= max_per_img and idx >= len(own) - n_flips:
                reason = 'tail'
            rows.append(dict(row=int(idx), box=own[idx, :4].tolist(), score=float(own[idx, 4]), reason=reason,
                             iou_margin=margin))
            out['unexplained'] += reason == 'unexplained'
    explain(ref, got, only_ref, only_got, out['only_ref'])
    explain(got, ref, only_got, only_ref, out['only_got'])
    return out


def mask_iou_of_pairs(ref_rles, got_rles, pairs):
    """The FINAL binary masks (decoded COCO RLE, what ``dt_isegmaps_rle`` carries) of every matched pair:
    (IoU [n], differing pixels [n])."""
    from . import rle
    iou, diff = np.ones(len(pairs)), np.zeros(len(pairs), np.int64)
    for k, (i, j) in enumerate(pairs):
        if ref_rles[i] == got_rles[j]:
            continue
        a, b = rle.decode(ref_rles[i]).astype(bool), rle.decode(got_rles[j]).astype(bool)
        union = np.logical_or(a, b).sum()
        diff[k] = int(np.logical_xor(a, b).sum())
        if union:
            iou[k] = np.logical_and(a, b).sum() / union
    return iou, diff
