"""Utterance sharding over the GPUs of one node.

Utterances are independent (each starts from [(0.0, [BOS], "")],
scripts/ns_utils.py:214) and the tables are read-only, so every rank decodes a
contiguous block with its own replica of the tables and the only exchange is
one final gather of token ids and lengths.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, end) of the contiguous block of rank `rank`: ceil(n/world) each, the tail may be short or empty."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def gather_results(ids: torch.Tensor, lens: torch.Tensor, n_total: int, dst: int = 0
                   ) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Collect every rank's (ids u8 [n_r, T+1], lens i32 [n_r]) on `dst`, in utterance order.

    One padded all_gather of each tensor (RCCL on GPUs, gloo on CPU); returns
    (ids [n_total, T+1], lens [n_total]) on dst and None elsewhere.  Without an
    initialised process group the inputs are returned unchanged.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return ids, lens
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    width = ids.shape[1]
    pad_ids = torch.zeros(per, width, dtype=ids.dtype, device=ids.device)
    pad_len = torch.zeros(per, dtype=lens.dtype, device=lens.device)
    n = ids.shape[0]
    pad_ids[:n] = ids
    pad_len[:n] = lens
    all_ids = torch.empty(world * per, width, dtype=ids.dtype, device=ids.device)
    all_len = torch.empty(world * per, dtype=lens.dtype, device=lens.device)
    dist.all_gather_into_tensor(all_ids, pad_ids)
    dist.all_gather_into_tensor(all_len, pad_len)
    if rank != dst:
        return None
    keep: List[int] = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        keep.extend(range(r * per, r * per + (hi - lo)))
    idx = torch.tensor(keep, dtype=torch.long, device=ids.device)
    return all_ids.index_select(0, idx), all_len.index_select(0, idx)
