"""Next-character prior of the n-gram tables and the training regulariser built on it.

Host side of nsd_cond_dist / nsd_lm_prior_ce / nsd_lm_prior_ce_grad: same names and argument meaning as
the reference's `CharNgramLM.cond_dist` (scripts/ns_utils.py:121-136, train_emg_llm.py:88-107) and
`ns_lm_regularizer` (train_emg_llm.py:254-316), batched and on the GPU.  No CPU fall-back.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

from . import lib as _lib
from .decoder import DeviceTables, _require_cuda, _stream_ptr
from .tables import BASE_CHARS, HostTrie, compile_ngram, flatten_trie

_CHAR_ID = {c: i for i, c in enumerate(BASE_CHARS)}


def tables_from_lm(lm, floor_vsz: bool = True) -> DeviceTables:
    """Device tables holding only an LM (the lexicon is left empty): what the prior needs.
    `lm` exposes order/counts/context/V like CharNgramLM; both copies of cond_dist use max(1, |V|)."""
    return DeviceTables(flatten_trie(HostTrie()), compile_ngram(lm, floor_vsz))


def cond_dist_batch(tables: DeviceTables, contexts: Sequence[str], form: str = "U") -> torch.Tensor:
    """P(next char | context) over BASE_CHARS for each context string: float32 [N, 37] on the device.

    form "U" follows scripts/ns_utils.py:111-136, form "T" train_emg_llm.py:88-107 (they differ at order 1).
    Contexts are lower-cased as the reference does; a character outside BASE_CHARS matches no count.
    """
    _require_cuda()
    if form not in _lib.PRIOR_FORMS:
        raise ValueError("form must be 'U' or 'T'")
    if tables.order < 1:
        raise ValueError("the tables hold no n-gram LM")
    N = len(contexts)
    keep = max(tables.order - 1, 1)                      # only the last order-1 characters matter
    ids = np.full((max(N, 1), keep), 0xFF, np.uint8)
    lens = np.zeros(max(N, 1), np.int32)
    for r, c in enumerate(contexts):
        tail = (c or "").lower()[-keep:]
        lens[r] = len(tail)
        ids[r, :len(tail)] = [_CHAR_ID.get(ch, 0xFF) for ch in tail]
    dev = torch.device("cuda", torch.cuda.current_device())
    d_ids, d_len = torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)
    out = torch.empty(N, len(BASE_CHARS), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().nsd_cond_dist(tables.handle, d_ids.data_ptr(), d_len.data_ptr(), N, keep,
                                         _lib.PRIOR_FORMS[form], out.data_ptr(), _stream_ptr()), "nsd_cond_dist")
    return out


class _PriorCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, out_ids, tables, form):
        B, L, V = logits.shape
        x = logits.detach().contiguous().float()
        ids = out_ids.detach().contiguous().to(torch.int64)
        pos = torch.empty(max(B * L, 1), dtype=torch.float32, device=x.device)
        stats = torch.empty(3, dtype=torch.float64, device=x.device)
        _lib.check(_lib.load().nsd_lm_prior_ce(tables.handle, x.data_ptr(), ids.data_ptr(), B, L, V, form,
                                               pos.data_ptr(), stats.data_ptr(), _stream_ptr()), "nsd_lm_prior_ce")
        ctx.save_for_backward(x, ids, stats)
        ctx.tables, ctx.form, ctx.in_dtype = tables, form, logits.dtype
        return stats[2].to(torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        x, ids, stats = ctx.saved_tensors
        B, L, V = x.shape
        g = grad_out.detach().to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(x)
        _lib.check(_lib.load().nsd_lm_prior_ce_grad(ctx.tables.handle, x.data_ptr(), ids.data_ptr(), B, L, V,
                                                    ctx.form, stats.data_ptr(), g.data_ptr(), grad.data_ptr(),
                                                    _stream_ptr()), "nsd_lm_prior_ce_grad")
        return grad.to(ctx.in_dtype), None, None, None


def ns_lm_regularizer(logits_ce: torch.Tensor, in_ids, out_ids: torch.Tensor, tables: DeviceTables,
                      form: str = "T") -> torch.Tensor:
    """Mean over the valid positions of CE(LM prior || softmax(logits_ce)) as a differentiable scalar
    (train_emg_llm.py:254-316): logits_ce [B, L, 40] on the device, out_ids [B, L] targets with -100 at the
    padding; `in_ids` is accepted and ignored like the reference does.  The reference returns 0 when no
    position is valid; so does this."""
    _require_cuda()
    if logits_ce.dim() != 3 or tuple(out_ids.shape) != tuple(logits_ce.shape[:2]):
        raise ValueError("logits_ce must be [B, L, V] and out_ids [B, L]")
    if not logits_ce.is_cuda or not out_ids.is_cuda:
        raise _lib.NsdError("ns_lm_regularizer needs device tensors (there is no CPU fall-back)")
    return _PriorCE.apply(logits_ce, out_ids, tables, _lib.PRIOR_FORMS[form])
