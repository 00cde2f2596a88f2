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


_KEPT: dict = {}


def _kept_rows(n_total: int, world: int, per: int, device) -> torch.Tensor:
    """Rows of the padded gather that hold utterances (the last shards may be short or empty); cached."""
    key = (n_total, world, str(device))
    idx = _KEPT.get(key)
    if idx is None:
        parts = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            parts.append(torch.arange(r * per, r * per + (hi - lo), dtype=torch.long))
        idx = torch.cat(parts).to(device)
        _KEPT.clear()
        _KEPT[key] = idx
    return idx


_BUFS: dict = {}


def _buffers(per: int, width: int, world: int, device):
    """Send / receive buffers of the gather, kept between calls (the padding rows stay zero)."""
    key = (per, width, world, str(device))
    b = _BUFS.get(key)
    if b is None:
        _BUFS.clear()
        b = (torch.zeros(per, width + 4, dtype=torch.uint8, device=device),
             torch.empty(world * per, width + 4, dtype=torch.uint8, device=device))
        _BUFS[key] = b
    return b


def gather_results(ids: torch.Tensor, lens: torch.Tensor, n_total: int, dst: int = 0, unpack: bool = True,
                   force: bool = False) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Collect every rank's (ids u8 [n_r, T+1], lens i32 [n_r]) on `dst`, in utterance order.

    One padded all_gather (RCCL on GPUs, gloo on CPU); returns
    (ids [n_total, T+1], lens [n_total]) on dst and None elsewhere.  Without an
    initialised process group, or with a group of one rank, the inputs are
    returned unchanged; ``force=True`` runs the collective even in a group of one
    rank (the result then comes out of the receive buffer like any other).  With
    ``unpack=False`` the exchange is enqueued and nothing is returned; `unpack_gathered`
    turns the receive buffer into the result later (a loop that only needs the
    last result does not pay for the others).
    """
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return ids, lens
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    width = ids.shape[1]
    n = ids.shape[0]
    # one collective: ids and the little-endian bytes of the int32 length travel in the same rows
    pad, out = _buffers(per, width, world, ids.device)
    if n < per:
        pad[n:].zero_()
    pad[:n, :width] = ids
    pad[:n, width:] = lens.to(torch.int32).contiguous().view(torch.uint8).reshape(n, 4)
    if ids.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the multi-rank path without RCCL (several ranks on one GPU): gloo exchanges host memory
        host = torch.empty(world * per, width + 4, dtype=torch.uint8)
        dist.all_gather_into_tensor(host, pad.cpu())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, pad)
    if not unpack or rank != dst:
        return None
    return unpack_gathered(out, n_total, width, lens.dtype)


def unpack_gathered(out: torch.Tensor, n_total: int, width: int, len_dtype=torch.int32
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(ids [n_total, width], lens [n_total]) from the receive buffer of `gather_results`."""
    world = dist.get_world_size()
    per = (n_total + world - 1) // world
    if per * world == n_total:
        rows = out                                  # equal shards: the gathered rows are already in order
    else:
        rows = out.index_select(0, _kept_rows(n_total, world, per, out.device))
    len_bytes = torch.empty(rows.shape[0], 4, dtype=torch.uint8, device=out.device)   # fresh, aligned storage
    len_bytes.copy_(rows[:, width:])
    all_len = len_bytes.view(torch.int32).reshape(-1)
    return rows[:, :width].contiguous(), all_len.to(len_dtype)


def last_gathered(per: int, width: int, world: int, device) -> torch.Tensor:
    """The receive buffer the last `gather_results` call with these sizes filled."""
    return _buffers(per, width, world, device)[1]
