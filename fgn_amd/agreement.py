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
