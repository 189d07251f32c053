"""Episode-level data parallelism: one process per GPU, episodes sharded round-robin,
detections gathered with ONE collective of fixed-size padded buffers per step.

The reference is single-process/single-GPU (main.py:365) and has no collective; episodes
(one query + its N*K supports) are independent, so the only exchange is
the gather of results.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests.  A detection record is [x1, y1, x2, y2, score, label] followed by the M x M mask
probabilities of that detection (the mask-head output the reference pastes and encodes,
fgn_roi_head.py:668-671; 14 x 14 at the reference's settings), plus a per-episode valid count:
~81 KB per episode, i.e. latency-bound - one all-gather per step, never one per episode.
The receiving rank turns any gathered episode into the reference's result dict
(``results_from_gathered``: paste + threshold + COCO RLE of the gathered probabilities with the
same fused kernel the producing rank runs, so the strings are byte-identical).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

REC_BOX = 6          # x1, y1, x2, y2, score, label


def shard_episodes(n_episodes: int, rank: int, world: int) -> List[int]:
    """Episode e runs on rank e mod world.  Ranks get unequal counts when world does not divide
    n_episodes; ``pack_detections(..., pad_to=episodes_per_rank(...))`` equalises the messages."""
    return list(range(rank, n_episodes, world))


def episodes_per_rank(n_episodes: int, world: int) -> int:
    """Message size (in episodes) every rank must send so that one fixed-size all-gather serves all."""
    return (n_episodes + world - 1) // world


def pack_detections(dets: list, max_det: int, pad_to: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """List of per-image device dicts (FGN.detect_device) -> (records [E, max_det, 6 + M*M], counts [E] int32).
    ``pad_to`` > len(dets) appends zero-count episodes (a rank with one episode fewer than the others)."""
    rows = []
    for d in dets:
        rec = torch.cat([d['det_bboxes'][:max_det], d['det_labels'][:max_det, None].float()], 1)
        if 'mask_prob' in d and d['mask_prob'] is not None:
            rec = torch.cat([rec, d['mask_prob'][:max_det].reshape(rec.shape[0], -1)], 1)
        rows.append(rec)
    recs = torch.stack(rows)
    cnts = torch.cat([d['n_dets'].reshape(1) for d in dets]).to(torch.int32)
    if pad_to is not None and pad_to > len(dets):
        extra = pad_to - len(dets)
        recs = torch.cat([recs, recs.new_zeros((extra,) + tuple(recs.shape[1:]))])
        cnts = torch.cat([cnts, cnts.new_zeros(extra)])
    return recs.contiguous(), cnts.contiguous()


def gather_detections(recs: torch.Tensor, cnts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks receive every rank's records: ([world, E, max_det, F], [world, E]).  Every rank must
    pass the same E (see ``episodes_per_rank``)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return recs[None], cnts[None]
    world = dist.get_world_size()
    # one message: counts ride along as an extra column of the flattened record tensor
    e, m, f = recs.shape
    msg = torch.cat([recs.reshape(e, m * f), cnts.to(recs.dtype)[:, None]], 1).contiguous()
    if dist.get_backend() == 'gloo' and msg.is_cuda:      # CPU rehearsals of the multi-rank flow: gloo has no CUDA all-gather
        host = msg.cpu()
        out = torch.empty((world * e, msg.shape[1]), dtype=msg.dtype)
        dist.all_gather_into_tensor(out, host)
        out = out.to(msg.device)
    else:
        out = torch.empty((world * e, msg.shape[1]), dtype=msg.dtype, device=msg.device)   # concatenated form
        dist.all_gather_into_tensor(out, msg)
    out = out.view(world, e, -1)
    return out[:, :, :m * f].reshape(world, e, m, f), out[:, :, m * f].round().to(torch.int32)


def interleave(gathered: torch.Tensor) -> torch.Tensor:
    """[world, E_local, ...] -> [world*E_local, ...] in global episode order (e = i*world + rank)."""
    return gathered.transpose(0, 1).reshape((-1,) + tuple(gathered.shape[2:]))


def results_from_gathered(recs: torch.Tensor, cnts: torch.Tensor, img_hw, mask_thr: float = 0.5,
                          rle_fn=None, skip_empty: bool = True) -> List[dict]:
    """Gathered records of E episodes ([E, max_det, 6 + M*M], [E]) -> the reference's result dicts
    (fgn.py:276-281: ``dt_scores``, ``dt_bboxes`` YXYX, ``dt_cat_ids``, ``dt_isegmaps_rle``).
    ``img_hw``: (H, W) or a list of E such pairs.  ``rle_fn(prob [n,M,M], boxes [n,5], H, W, thr) ->
    list of RLE dicts``; default = the fused HIP paste+RLE kernel on the records' device (with the host
    encoder only for strings that overflow the device caps), which needs a GPU; ``skip_empty`` = the paste
    semantics of the producing detector (``FGN.paste_semantics == 'cpu'``)."""
    e, max_det, f = recs.shape
    m = int(round((f - REC_BOX) ** 0.5))
    if f <= REC_BOX or m * m != f - REC_BOX:
        raise ValueError('records carry no mask probabilities')
    counts = cnts.cpu().numpy()
    out = []
    for i in range(e):
        n = int(counts[i])
        h, w = img_hw[i] if isinstance(img_hw[0], (tuple, list)) else img_hw
        rec = recs[i, :n]
        prob = rec[:, REC_BOX:].reshape(n, m, m).contiguous()
        boxes = rec[:, :5].contiguous()
        rles = (rle_fn(prob, boxes, int(h), int(w), mask_thr) if rle_fn else
                _hip_rle(prob, boxes, int(h), int(w), mask_thr, skip_empty)) if n else []
        b = boxes.cpu().numpy()
        out.append({'dt_scores': b[:, 4].copy(), 'dt_bboxes': b[:, [1, 0, 3, 2]].copy(),
                    'dt_cat_ids': rec[:, 5].round().to(torch.int64).cpu().numpy(), 'dt_isegmaps_rle': rles})
    return out


def _hip_rle(prob, boxes, h, w, thr, skip_empty=True):
    from . import ops, rle
    by, ln, ovf = ops.mask_rle(prob, boxes, h, w, thr, skip_empty=skip_empty)
    by, ln, ovf = by.cpu().numpy(), ln.cpu().numpy(), ovf.cpu().numpy()
    res = []
    for j in range(prob.shape[0]):
        if ovf[j]:
            dense = ops.mask_paste(prob[j:j + 1].contiguous(), boxes[j:j + 1].contiguous(), h, w, thr, skip_empty=skip_empty)
            res.append(rle.encode(dense[0].cpu().numpy()))
        else:
            res.append({'size': [h, w], 'counts': by[j, :ln[j]].tobytes()})
    return res


# ------------------------------------------------------------------------------------------
# data-parallel training: one episode batch per rank, gradients averaged with ONE all-reduce
# ------------------------------------------------------------------------------------------
def allreduce_mean(tensors: dict, keys=None) -> dict:
    """Average a dict of same-device fp32 tensors over the ranks with a single all-reduce of one flat bucket (the heads
    hold ~26 M parameters: one 105 MB message per step - bandwidth-bound on the xGMI ring, never one collective per
    tensor).  ``keys``: the rank-invariant key list of the bucket (default: the sorted keys of ``tensors``, which must
    then be the same set on every rank); every key must be present - a rank whose batch produced no gradient for a
    tensor passes zeros, never a shorter dict (ranks disagreeing on the bucket length hang or corrupt the collective).
    No process group or a single rank: the dict is returned unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    keys = sorted(tensors) if keys is None else list(keys)
    missing = [k for k in keys if k not in tensors]
    if missing:
        raise KeyError(f'allreduce_mean: no tensor for {missing[:3]} on this rank; fill missing gradients with zeros')
    if not keys:
        return tensors
    flat = torch.cat([tensors[k].reshape(-1) for k in keys])
    if flat.is_cuda and dist.get_backend() == 'gloo':        # CPU rehearsals of the multi-rank path: stage through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    out, o = {}, 0
    for k in keys:
        n = tensors[k].numel()
        out[k] = flat[o:o + n].view_as(tensors[k])
        o += n
    return out
