"""Episode-level data parallelism: one process per GPU, episodes sharded round-robin,
detections gathered with ONE collective of fixed-size padded buffers per step.

The reference is single-process/single-GPU (main.py:365) and has no collective; episodes
(one query + its N*K supports) are independent, so the only exchange is
the gather of results.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests.  A detection record is [x1, y1, x2, y2, score, label] and a per-episode valid
count; messages are ~2.4 KB per episode, i.e. latency-bound - one all-gather per step,
never one per episode.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_episodes(n_episodes: int, rank: int, world: int) -> List[int]:
    """Episode e runs on rank e mod world."""
    return list(range(rank, n_episodes, world))


def pack_detections(dets: list, max_det: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """List of per-image device dicts (FGN.detect_device) -> ([E,max_det,6], [E] int32)."""
    recs = torch.stack([torch.cat([d['det_bboxes'][:max_det], d['det_labels'][:max_det, None].float()], 1)
                        for d in dets])
    cnts = torch.cat([d['n_dets'] for d in dets]).to(torch.int32)
    return recs.contiguous(), cnts.contiguous()


def gather_detections(recs: torch.Tensor, cnts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """All ranks receive every rank's records: ([world,E,max_det,6], [world,E])."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return recs[None], cnts[None]
    world = dist.get_world_size()
    # one message: counts ride along as an extra row of the record tensor
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
