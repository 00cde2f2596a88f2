"""Host side of the MI355X NS decoder.

``constrained_beam_decode`` keeps the reference's signature and call pattern
(scripts/ns_utils.py:194-204; the in-file copies at
inference_emg_llm.py:222-231 and eval_ar.py:214-227 are selected with
``variant``), ``NSDecoder`` is the batched form: N utterances x B beams advance
in lock step, one ``nsd_step`` launch per step, state and back-pointers
resident in HBM.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from .text import finish_transcripts, ids_to_text  # noqa: F401  (ids_to_text: scripts/ns_utils.py:284-296)
from .tables import (BASE_CHARS, BOS_ID, EOS_ID, PAD_ID, VOCAB, HostTrie, NgramTables, compile_ngram,
                     flatten_trie)

__all__ = ["DeviceTables", "NSDecoder", "Workspace", "constrained_beam_decode", "ids_to_text", "clear_table_cache",
           "ctc_greedy_batch", "lm_logp_batch", "rerank_gather_batch", "log_softmax_rows", "rerank_lm_batch"]


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda():
    if not torch.cuda.is_available():
        raise _lib.NsdError("the NS decoder runs on an MI355X only: no HIP device is visible "
                            "(there is no CPU fall-back)")


class DeviceTables:
    """Trie + n-gram tables resident on the current device (immutable)."""

    def __init__(self, nodes: np.ndarray, ngram: Optional[NgramTables]):
        _require_cuda()
        L = _lib.load()
        self.nodes = np.ascontiguousarray(nodes)
        self.ngram = ngram
        trie = _lib.TrieCSR(_lib.np_ptr(self.nodes), int(self.nodes.shape[0]))
        lm = None
        if ngram is not None:
            self._g = np.ascontiguousarray(ngram.grams)
            self._c = np.ascontiguousarray(ngram.ctxs)
            self._cr = np.ascontiguousarray(ngram.ctx_recs)
            self._gr = np.ascontiguousarray(ngram.gram_recs)
            lm = _lib.NgramHash(ngram.order, ngram.init_ctx, self._g.shape[0], self._c.shape[0],
                                _lib.np_ptr(self._g), _lib.np_ptr(self._c), ngram.unseen,
                                self._cr.shape[0], self._gr.shape[0],
                                _lib.np_ptr(self._cr) if self._cr.shape[0] else None,
                                _lib.np_ptr(self._gr) if self._gr.shape[0] else None)
        h = C.c_void_p()
        _lib.check(L.nsd_tables_create(C.byref(trie), C.byref(lm) if lm is not None else None, C.byref(h)),
                   "nsd_tables_create")
        self.handle = h
        self.order = ngram.order if ngram is not None else 0

    @classmethod
    def from_objects(cls, trie, lm=None, floor_vsz: Optional[bool] = None) -> "DeviceTables":
        """From reference-style objects: a trie exposing .next/.end and an LM
        exposing .order/.counts/.context/.V (or None)."""
        for attr in ("next", "end"):
            if not hasattr(trie, attr):
                raise TypeError("trie must expose .next and .end like the reference Trie")
        ngram = None
        if lm is not None:
            for attr in ("order", "counts", "context", "V"):
                if not hasattr(lm, attr):
                    raise TypeError("lm must expose order/counts/context/V like CharNgramLM "
                                    "(KenLM models are not supported)")
            ngram = compile_ngram(lm, floor_vsz)
        return cls(flatten_trie(trie), ngram)

    @classmethod
    def from_words(cls, words: Sequence[str], lm=None, floor_vsz: Optional[bool] = None) -> "DeviceTables":
        return cls.from_objects(HostTrie.from_words(words), lm, floor_vsz)

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().nsd_tables_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Workspace:
    """Caller-owned device buffers of one batch (N utterances x B beams, T steps)."""

    def __init__(self, N: int, B: int, T: int, device, with_masks: bool = False):
        self.N, self.B, self.T = N, B, T
        NB = max(N * B, 1)
        kw = dict(device=device)
        self.score = torch.zeros(NB, dtype=torch.float64, **kw)
        self.lmsum = torch.zeros(NB, dtype=torch.float64, **kw)
        self.lmsum_e = torch.zeros(NB, dtype=torch.float64, **kw)
        self.ctx = torch.zeros(NB, dtype=torch.int64, **kw)
        self.hist = torch.zeros(NB, dtype=torch.int64, **kw)
        self.ctx_id = torch.zeros(NB, dtype=torch.int32, **kw)
        self.ctxe_id = torch.zeros(NB, dtype=torch.int32, **kw)
        self.node = torch.zeros(NB, dtype=torch.int32, **kw)
        self.meta = torch.zeros(NB, dtype=torch.int32, **kw)
        self.utt = torch.zeros(max(N, 1), dtype=torch.int32, **kw)
        self.done_count = torch.zeros(1, dtype=torch.int32, **kw)
        self.logits = torch.zeros(max(NB, 1), VOCAB, dtype=torch.float32, **kw)
        self.backptr = torch.zeros(max(T, 1), max(NB, 1), dtype=torch.uint8, **kw)
        self.token = torch.zeros(max(T, 1), max(NB, 1), dtype=torch.uint8, **kw)
        self.masks = torch.zeros(max(T, 1), max(NB, 1), dtype=torch.int64, **kw) if with_masks else None
        self.out_ids = torch.zeros(max(N, 1), T + 1, dtype=torch.uint8, **kw)
        self.out_len = torch.zeros(max(N, 1), dtype=torch.int32, **kw)
        self.out_score = torch.zeros(max(N, 1), dtype=torch.float64, **kw)
        self.ptrs = _lib.BeamStatePtrs(self.score.data_ptr(), self.lmsum.data_ptr(), self.lmsum_e.data_ptr(),
                                       self.ctx.data_ptr(), self.hist.data_ptr(), self.ctx_id.data_ptr(),
                                       self.ctxe_id.data_ptr(), self.node.data_ptr(), self.meta.data_ptr(),
                                       self.utt.data_ptr(), self.done_count.data_ptr())
        self.steps_run = 0


class NSDecoder:
    """Batched NS beam decoder; parameters as in the reference signatures."""

    def __init__(self, tables: DeviceTables, beam: int, max_new_tokens: int, *, variant: str = "U",
                 lm_weight: float = 0.3, oov_penalty: float = 1.0, allow_eos_anytime: bool = True,
                 word_bonus: float = 0.0, length_penalty_alpha: float = 0.6, min_len_for_eos: int = 8,
                 use_lm: Optional[bool] = None, device=None, lanes_per_utt: int = 0, lds_trie_nodes: int = 0):
        _require_cuda()
        if variant not in _lib.VARIANTS:
            raise ValueError("variant must be 'U', 'I' or 'E'")
        if not 1 <= int(beam) <= _lib.MAX_BEAM:
            raise ValueError(f"beam {beam} outside [1, {_lib.MAX_BEAM}]")
        if not 0 <= int(max_new_tokens) <= _lib.MAX_STEPS:
            raise ValueError(f"max_new_tokens {max_new_tokens} outside [0, {_lib.MAX_STEPS}]")
        self.L = _lib.load()
        self.tables = tables
        self.B, self.T, self.variant = int(beam), int(max_new_tokens), variant
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if use_lm is None:
            use_lm = tables.order > 0
        self.params = _lib.make_params(variant, use_lm=use_lm and tables.order > 0, beta=lm_weight,
                                       gamma=oov_penalty, kappa=word_bonus, alpha=length_penalty_alpha,
                                       min_len_for_eos=min_len_for_eos, allow_eos_anytime=allow_eos_anytime,
                                       lanes_per_utt=lanes_per_utt, lds_trie_nodes=lds_trie_nodes)

    # -- low level ---------------------------------------------------------
    def workspace(self, N: int, with_masks: bool = False) -> Workspace:
        return Workspace(int(N), self.B, self.T, self.device, with_masks)

    def init(self, ws: Workspace) -> None:
        _lib.check(self.L.nsd_state_init(self.tables.handle, C.byref(ws.ptrs), ws.N, ws.B, _stream_ptr()),
                   "nsd_state_init")
        ws.steps_run = 0

    def step(self, ws: Workspace, logits: Optional[torch.Tensor] = None,
             mask_out: Optional[torch.Tensor] = None) -> None:
        """One scorer launch on step index ws.steps_run.  ``mask_out`` (int64 [N*B], device) receives the lexicon
        masks of this step's live hypotheses instead of the workspace's per-step mask array."""
        t = ws.steps_run
        if t >= ws.T:
            raise ValueError("all max_new_tokens steps already ran")
        lg = ws.logits if logits is None else logits
        if lg.dtype != torch.float32 or not lg.is_contiguous() or lg.numel() != max(ws.N * ws.B, 1) * VOCAB \
                or lg.device != ws.score.device:
            raise ValueError("logits must be a contiguous float32 [N*B, 40] tensor on the decoder's device")
        NB = max(ws.N * ws.B, 1)
        mask_ptr = ws.masks.data_ptr() + t * NB * 8 if ws.masks is not None else None
        if mask_out is not None:
            if mask_out.dtype != torch.int64 or mask_out.numel() < NB or not mask_out.is_contiguous() \
                    or mask_out.device != ws.score.device:
                raise ValueError("mask_out must be a contiguous int64 tensor of N*B elements on the decoder's device")
            mask_ptr = mask_out.data_ptr()
        _lib.check(self.L.nsd_step(self.tables.handle, C.byref(self.params), lg.data_ptr(),
                                   C.byref(ws.ptrs), C.byref(ws.ptrs),
                                   ws.backptr.data_ptr() + t * NB, ws.token.data_ptr() + t * NB, mask_ptr,
                                   ws.N, ws.B, VOCAB, _stream_ptr()), "nsd_step")
        ws.steps_run = t + 1

    def all_done(self, ws: Workspace) -> bool:
        return int(ws.done_count.item()) >= ws.N

    def gather_ids(self, ws: Workspace):
        """Token histories of all current hypotheses: (ids u8 [N*B, T+1], len i32 [N*B])."""
        ids = torch.zeros(max(ws.N * ws.B, 1), ws.T + 1, dtype=torch.uint8, device=self.device)
        lens = torch.zeros(max(ws.N * ws.B, 1), dtype=torch.int32, device=self.device)
        _lib.check(self.L.nsd_gather_ids(C.byref(ws.ptrs), ws.backptr.data_ptr(), ws.token.data_ptr(),
                                         ws.T, ws.N, ws.B, ids.data_ptr(), lens.data_ptr(), _stream_ptr()),
                   "nsd_gather_ids")
        return ids, lens

    def finish(self, ws: Workspace):
        """Back-trace the best hypothesis of every utterance; raises like
        torch.topk when a fall-back needed more candidates than the vocabulary
        holds (variants I/E)."""
        _lib.check(self.L.nsd_backtrace(C.byref(ws.ptrs), ws.backptr.data_ptr(), ws.token.data_ptr(),
                                        ws.T, ws.N, ws.B, ws.out_ids.data_ptr(), ws.out_len.data_ptr(),
                                        ws.out_score.data_ptr(), _stream_ptr()), "nsd_backtrace")
        self.raise_on_topk_error(ws)
        return ws.out_ids[:ws.N], ws.out_len[:ws.N], ws.out_score[:ws.N]

    # -- drivers -----------------------------------------------------------
    def decode_batch(self, step_fn_batched: Callable, N: int, *, needs_ids: bool = True, check_every: int = 8,
                     ws: Optional[Workspace] = None):
        """``step_fn_batched(ids_u8[N*B, T+1], len_i32[N*B], ws) -> float32 logits [N*B, 40]``
        on the decoder's device (rows of finished / absent hypotheses are ignored)."""
        ws = ws or self.workspace(N)
        self.init(ws)
        for t in range(self.T):
            ids, lens = self.gather_ids(ws) if needs_ids else (None, None)
            logits = step_fn_batched(ids, lens, ws)
            self.step(ws, logits)
            if check_every and (t + 1) % check_every == 0 and self.all_done(ws):
                break
        return self.finish(ws)

    def synth_logits(self, ws: Workspace, base_seed: int, utt0: int, targets: torch.Tensor,
                     target_len: torch.Tensor, margin: float) -> None:
        _lib.check(self.L.nsd_synth_logits(C.byref(ws.ptrs), base_seed, utt0, targets.data_ptr(),
                                           target_len.data_ptr(), targets.shape[1], float(margin),
                                           ws.logits.data_ptr(), ws.N, ws.B, _stream_ptr()), "nsd_synth_logits")

    def decode_synth(self, targets: torch.Tensor, target_len: torch.Tensor, *, base_seed: int, margin: float,
                     utt0: int = 0, check_every: int = 8, ws: Optional[Workspace] = None):
        """Decode with the synthetic integer-hash AR head (int8 targets [N, S] on device); the step loop is
        driven from Python, one launch at a time (this is the form that can record lexicon masks)."""
        N = targets.shape[0]
        ws = ws or self.workspace(N)
        self.init(ws)
        for t in range(self.T):
            self.synth_logits(ws, base_seed, utt0, targets, target_len, margin)
            self.step(ws)
            if check_every and (t + 1) % check_every == 0 and self.all_done(ws):
                break
        return self.finish(ws)

    def decode_synth_native(self, targets: torch.Tensor, target_len: torch.Tensor, *, base_seed: int,
                            margin: float, utt0: int = 0, check_every: int = 0, ws: Optional[Workspace] = None,
                            step_ms: Optional[np.ndarray] = None):
        """The same decode with the whole step loop enqueued by native code (nsd_decode_synth).  ``step_ms``
        (float32 [T], host) receives the HIP-event duration of every nsd_step launch."""
        N = targets.shape[0]
        ws = ws or self.workspace(N)
        ran = C.c_int32(0)
        if step_ms is not None and (step_ms.dtype != np.float32 or step_ms.shape[0] < self.T):
            raise ValueError("step_ms must be float32 with at least max_new_tokens entries")
        _lib.check(self.L.nsd_decode_synth(
            self.tables.handle, C.byref(self.params), C.byref(ws.ptrs), base_seed, utt0, targets.data_ptr(),
            target_len.data_ptr(), targets.shape[1], float(margin), ws.logits.data_ptr(), ws.backptr.data_ptr(),
            ws.token.data_ptr(), ws.T, ws.N, ws.B, ws.out_ids.data_ptr(), ws.out_len.data_ptr(),
            ws.out_score.data_ptr(), int(check_every), step_ms.ctypes.data if step_ms is not None else None,
            C.byref(ran), _stream_ptr()), "nsd_decode_synth")
        ws.steps_run = ran.value
        return ws.out_ids[:ws.N], ws.out_len[:ws.N], ws.out_score[:ws.N]

    def raise_on_topk_error(self, ws: Workspace) -> None:
        if not ws.N:
            return
        bad = int((ws.utt[:ws.N] & (_lib.U_ERR_TOPK | _lib.U_ERR_INTERNAL)).max().item())
        if bad & _lib.U_ERR_INTERNAL:
            raise _lib.NsdError("nsd_step reported an overflowing candidate buffer (a defect of the library)")
        if bad & _lib.U_ERR_TOPK:
            raise RuntimeError("selected index k out of range (beam larger than the vocabulary in the "
                               "top-k fall-back)")


# ----------------------------------------------------------------------
# drop-in for the reference function
# ----------------------------------------------------------------------
_TABLE_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()


_DIGEST_LIMIT = 1 << 20          # dictionary entries / trie nodes a digest looks at


def _fingerprint(trie, lm) -> tuple:
    """Digest of the objects' contents, so that an `lm.fit(...)` or `trie.insert(...)` on an object the cache
    already knows is noticed even when no size changes: the n-gram and context counts are summed, the trie's nodes
    and word ends are counted (each over at most _DIGEST_LIMIT entries; beyond that `clear_table_cache()` is the
    sure way).  Objects without the reference's attributes get a constant: building the tables then raises the
    TypeError that names what is missing."""
    from itertools import islice
    nodes = ends = 0
    stack = [trie]
    while stack and nodes < _DIGEST_LIMIT:
        nd = stack.pop()
        nxt = getattr(nd, "next", None)
        if not isinstance(nxt, dict):
            break
        nodes += 1
        ends += 1 if getattr(nd, "end", False) else 0
        stack.extend(nxt.values())
    t = (nodes, ends)
    if lm is None:
        return t, None
    counts, context = getattr(lm, "counts", None), getattr(lm, "context", None)
    if not isinstance(counts, dict) or not isinstance(context, dict):
        return t, ("?",)
    return t, (int(getattr(lm, "order", 0)), len(counts), len(context), len(getattr(lm, "V", ())),
               sum(islice(counts.values(), _DIGEST_LIMIT)), sum(islice(context.values(), _DIGEST_LIMIT)))


def clear_table_cache() -> None:
    """Forget the device tables built for earlier (trie, lm) objects (and the one-utterance workspaces)."""
    _TABLE_CACHE.clear()
    _WS_CACHE.clear()


_WS_CACHE: "OrderedDict[tuple, Workspace]" = OrderedDict()


_HOST_BUFS: dict = {}


def _host_buffers(B: int):
    """Pinned host buffers of the drop-in entry point for beam B: logits rows [B, 40] and the step's outcome."""
    b = _HOST_BUFS.get(B)
    if b is None:
        if len(_HOST_BUFS) > 8:
            _HOST_BUFS.clear()
        b = _HOST_BUFS[B] = (torch.zeros(B, VOCAB, dtype=torch.float32).pin_memory(),
                             torch.zeros(2 * B + 4, dtype=torch.uint8).pin_memory())
    return b


def _workspace_for(dec: "NSDecoder") -> "Workspace":
    """One-utterance workspace of the drop-in entry point, kept per (beam, steps, device): a call per utterance
    re-initialises its 18 device buffers instead of allocating them again."""
    key = (dec.B, dec.T, str(dec.device))
    ws = _WS_CACHE.get(key)
    if ws is None:
        ws = _WS_CACHE[key] = dec.workspace(1)
        while len(_WS_CACHE) > 4:
            _WS_CACHE.popitem(last=False)
    else:
        _WS_CACHE.move_to_end(key)
    return ws


def _tables_for(trie, lm, floor_vsz) -> DeviceTables:
    key = (id(trie), id(lm), floor_vsz, torch.cuda.current_device(), _fingerprint(trie, lm))
    hit = _TABLE_CACHE.get(key)
    if hit is not None and hit[0] is trie and hit[1] is lm:
        _TABLE_CACHE.move_to_end(key)
        return hit[2]
    t = DeviceTables.from_objects(trie, lm, floor_vsz)
    _TABLE_CACHE[key] = (trie, lm, t)
    while len(_TABLE_CACHE) > 4:
        _TABLE_CACHE.popitem(last=False)
    return t


def _check_tt(TT) -> None:
    TT.BASE_CHARS.index(" ")           # ValueError without a space, as the reference (ns_utils.py:215)
    if (TT.BASE_CHARS != BASE_CHARS or TT.VOCAB_OFFSET != 0 or TT.BOS_IDX != BOS_ID or TT.EOS_IDX != EOS_ID
            or TT.PAD_IDX != PAD_ID or TT.VOCAB_SIZE != VOCAB):
        raise NotImplementedError("the HIP decoder implements the reference TextTransform layout only "
                                  "(a-z0-9 + space, BOS=37, EOS=38, PAD=39, offset 0)")


def constrained_beam_decode(step_fn, TT, beam: int, max_new_tokens: int, trie, lm=None,
                            lm_weight: float = 0.3, oov_penalty: float = 1.0, allow_eos_anytime: bool = True,
                            *, variant: str = "U", word_bonus: float = 0.0,
                            length_penalty_alpha: float = 0.6, min_len_for_eos: int = 8) -> List[int]:
    """Drop-in for the reference's single-utterance decoder.

    ``step_fn(ids: List[int]) -> Tensor[1, V]`` is called once per live
    hypothesis per step, in rank order, exactly as the reference calls it;
    everything between those calls (log-softmax, lexicon mask, LM fusion,
    pruning) runs in one nsd_step launch.  variant "U" = scripts/ns_utils.py,
    "I" = inference_emg_llm.py, "E" = eval_ar.py (keyword-only extras).
    """
    _require_cuda()
    _check_tt(TT)
    tables = _tables_for(trie, lm, None)
    dec = NSDecoder(tables, beam, max_new_tokens, variant=variant, lm_weight=lm_weight,
                    oov_penalty=oov_penalty, allow_eos_anytime=allow_eos_anytime, word_bonus=word_bonus,
                    length_penalty_alpha=length_penalty_alpha, min_len_for_eos=min_len_for_eos,
                    use_lm=lm is not None)
    B = dec.B
    ws = _workspace_for(dec)
    dec.init(ws)
    hyps: List[List[int]] = [[BOS_ID]]
    # torch.topk(k=beam) of the reference's fall-back raises when beam exceeds the vocabulary (copies I and E,
    # inference_emg_llm.py:281): that happens right after step_fn was called for the first hypothesis without a
    # lexicon-consistent option, so the remaining hypotheses of the step are never evaluated
    raises_on_fallback = B > VOCAB and variant != "U"
    # Host side of a step: rows that step_fn returns on the host are collected in one pinned buffer and cross to the
    # device in one copy (rows it returns on a device are copied there directly); the step's outcome (utterance word,
    # back-pointers, tokens) comes back in one read.  Rows of hypotheses that are not live are never read by the
    # kernel, so the buffers keep whatever they held.
    rows, back = _host_buffers(B)
    rows_np = rows.numpy()
    logits = ws.logits
    for t in range(dec.T):
        dev_rows = []
        any_host = False
        if raises_on_fallback:
            meta_h = ws.meta[:len(hyps)].tolist()
            node_h = ws.node[:len(hyps)].tolist()
        for r, ids in enumerate(hyps):
            if ids[-1] == EOS_ID:
                continue
            out = step_fn(list(ids))
            if raises_on_fallback:
                dead = (meta_h[r] & _lib.F_DEAD) or (node_h[r] & 0xFFFFFFFF) == 0xFFFFFFFF
                mask = 0 if dead else int(tables.nodes["mask"][node_h[r]])
                if mask & ((1 << 37) - 1) == 0:
                    raise RuntimeError("selected index k out of range")
            v = out.detach().reshape(-1)
            if v.device.type == "cpu":
                rows_np[r] = v.numpy() if v.dtype == torch.float32 else v.to(torch.float32).numpy()
                any_host = True
            else:
                dev_rows.append((r, v))
        if any_host:
            logits.copy_(rows, non_blocking=True)
        for r, v in dev_rows:
            logits[r].copy_(v)
        dec.step(ws, logits)
        back.copy_(torch.cat((ws.backptr[t, :B], ws.token[t, :B], ws.utt[:1].view(torch.uint8))))     # one read, synchronises
        raw = back.numpy()
        um = int.from_bytes(raw[2 * B:2 * B + 4].tobytes(), "little")
        if um & _lib.U_ERR_INTERNAL:
            raise _lib.NsdError("nsd_step reported an overflowing candidate buffer (a defect of the library)")
        if um & _lib.U_ERR_TOPK:
            raise RuntimeError("selected index k out of range")
        nb = um & 0xFF
        par = raw[:nb].tolist()
        tok = raw[B:B + nb].tolist()
        hyps = [hyps[p] if k == _lib.TOKEN_CARRY else hyps[p] + [k] for p, k in zip(par, tok)]
        if um & _lib.U_DONE:
            break
    return hyps[0]


def transcribe_batch(TT, ar_ids: torch.Tensor, ar_len: torch.Tensor, tables: "DeviceTables", is_word,
                     lex_words: Sequence[str], ctc_logits: Optional[torch.Tensor] = None) -> List[str]:
    """Final transcripts of a decoded batch as the driver produces them (inference_emg_llm.py:436-464): AR ids ->
    text -> postprocess; with ``ctc_logits`` [N, T', 40] also the CTC greedy proposal (on the GPU) and the LM
    choice between the two.  The LM scores come from the same tables the decoder used."""
    from .tables import host_logp
    ctc_ids = ctc_len = None
    logp = None
    if ctc_logits is not None:
        if tables.ngram is None:
            raise ValueError("the CTC rerank needs n-gram tables")
        ci, cl = ctc_greedy_batch(ctc_logits)
        ctc_ids, ctc_len = ci.cpu().numpy(), cl.cpu().numpy()
        logp = lambda s: host_logp(tables.ngram, s)          # noqa: E731
    return finish_transcripts(TT, ar_ids.cpu().numpy(), ar_len.cpu().numpy(), is_word, lex_words, logp,
                              ctc_ids, ctc_len)


# ----------------------------------------------------------------------
# CTC proposal and rerank pieces
# ----------------------------------------------------------------------
def ctc_greedy_batch(logits: torch.Tensor):
    """(N, T', 40) float32 device logits -> (ids u8 [N, T'], len i32 [N])
    (inference_emg_llm.py:445-455)."""
    _require_cuda()
    N, Tp, V = logits.shape
    if logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("logits must be contiguous float32")
    ids = torch.zeros(max(N, 1), max(Tp, 1), dtype=torch.uint8, device=logits.device)
    lens = torch.zeros(max(N, 1), dtype=torch.int32, device=logits.device)
    _lib.check(_lib.load().nsd_ctc_greedy(logits.data_ptr(), N, Tp, V, ids.data_ptr(), lens.data_ptr(),
                                          _stream_ptr()), "nsd_ctc_greedy")
    return ids[:N], lens[:N]


def lm_logp_batch(tables: DeviceTables, text_ids: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """CharNgramLM.logp of N strings given as base-character ids u8 [N, S]."""
    _require_cuda()
    N, S = text_ids.shape
    out = torch.zeros(max(N, 1), dtype=torch.float64, device=text_ids.device)
    _lib.check(_lib.load().nsd_lm_logp(tables.handle, text_ids.data_ptr(), lens.data_ptr(), N, S,
                                       out.data_ptr(), _stream_ptr()), "nsd_lm_logp")
    return out[:N]


def rerank_lm_batch(tables: DeviceTables, ar_ids: torch.Tensor, ar_len: torch.Tensor,
                    ctc_ids: torch.Tensor, ctc_len: torch.Tensor):
    """AR hypothesis vs CTC proposal by lm.logp(s) - 0.002*len(s) (inference_emg_llm.py:461-464, texts as
    ids_to_text gives them, no postprocess): returns (choice u8 [N] with 1 = CTC, scores f64 [N, 2])."""
    _require_cuda()
    N = ar_ids.shape[0]
    choice = torch.zeros(max(N, 1), dtype=torch.uint8, device=ar_ids.device)
    scores = torch.zeros(max(N, 1), 2, dtype=torch.float64, device=ar_ids.device)
    _lib.check(_lib.load().nsd_rerank_lm(tables.handle, ar_ids.data_ptr(), ar_len.data_ptr(), ar_ids.shape[1],
                                         ctc_ids.data_ptr(), ctc_len.data_ptr(), ctc_ids.shape[1], N,
                                         choice.data_ptr(), scores.data_ptr(), _stream_ptr()), "nsd_rerank_lm")
    return choice[:N], scores[:N]


def rerank_gather_batch(logits: torch.Tensor, target: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    """Teacher-forced sum of token log-probs (eval_ar.py:574-589):
    logits float32 [N, L, 40], target u8 [N, L], lens i32 [N] -> f64 [N]."""
    _require_cuda()
    N, Lm, V = logits.shape
    out = torch.zeros(max(N, 1), dtype=torch.float64, device=logits.device)
    _lib.check(_lib.load().nsd_rerank_gather(logits.data_ptr(), target.data_ptr(), lens.data_ptr(), N, Lm, V,
                                             out.data_ptr(), _stream_ptr()), "nsd_rerank_gather")
    return out[:N]


def transcribe_batch_joint(TT, ar_ids: torch.Tensor, ar_len: torch.Tensor, tables: "DeviceTables", is_word,
                           lex_words: Sequence[str], ctc_logits: torch.Tensor, teacher_logits: Callable,
                           lm_weight: float):
    """The tail of eval_ar.py for a batch (eval_ar.py:546-596): AR and CTC transcripts, post-processed, then the
    joint choice ``ar_logprob(s) + lm_weight * lm.logp(s)`` with the AR transcript winning ties.

    ``teacher_logits(seq u8 [M, L], lens i32 [M]) -> float32 [M, L-1, 40]`` on the device: row i of sequence m
    holds the logits after the prefix ``seq[m, :i+1]`` (what ``step_fn(seq[:i+1])`` returns), for the 2N
    sequences BOS + characters + EOS of the candidates.  Returns (transcripts, scores f64 [N, 2])."""
    from .tables import host_logp
    ci, cl = ctc_greedy_batch(ctc_logits)
    ci, cl = ci.cpu().numpy(), cl.cpu().numpy()
    ar_np, al = ar_ids.cpu().numpy(), ar_len.cpu().numpy()
    N = len(al)
    hyp_ar = finish_transcripts(TT, ar_np, al, is_word, lex_words)
    wrapped = [[TT.BOS_IDX] + [int(t) for t in ci[u][:int(cl[u])]] + [TT.EOS_IDX] for u in range(N)]
    hyp_ctc = finish_transcripts(TT, wrapped, [len(w) for w in wrapped], is_word, lex_words)
    cands = hyp_ar + hyp_ctc
    seqs = [[TT.BOS_IDX] + [TT.VOCAB_OFFSET + TT.BASE_CHARS.index(c) for c in s if c in TT.BASE_CHARS] + [TT.EOS_IDX]
            for s in cands]
    L = max(len(q) for q in seqs)
    seq = np.full((2 * N, L), TT.PAD_IDX, np.uint8)
    lens = np.zeros(2 * N, np.int32)
    for m, q in enumerate(seqs):
        seq[m, :len(q)] = q
        lens[m] = len(q)
    dev = ctc_logits.device
    seq_d, lens_d = torch.from_numpy(seq).to(dev), torch.from_numpy(lens).to(dev)
    logits = teacher_logits(seq_d, lens_d)
    if logits.shape != (2 * N, L - 1, VOCAB) or logits.dtype != torch.float32:
        raise ValueError("teacher_logits must return float32 [2N, L-1, 40]")
    ar_lp = rerank_gather_batch(logits.contiguous(), seq_d[:, 1:].contiguous(), lens_d - 1).cpu().numpy()
    scores = np.zeros((N, 2), np.float64)
    out = []
    for u in range(N):
        for c, s in ((0, hyp_ar[u]), (1, hyp_ctc[u])):
            part = host_logp(tables.ngram, s) if tables.ngram is not None else 0.0
            scores[u, c] = (float(ar_lp[u + c * N]) if s else -1e9) + lm_weight * part      # eval_ar.py:575-576,591-593
        out.append(hyp_ctc[u] if scores[u, 1] > scores[u, 0] else hyp_ar[u])
    return out, scores


def log_softmax_rows(x: torch.Tensor) -> torch.Tensor:
    """The decoder's deterministic fp32 log-softmax over the last dim (=40)."""
    _require_cuda()
    rows = x.numel() // x.shape[-1]
    out = torch.empty_like(x)
    _lib.check(_lib.load().nsd_log_softmax(x.data_ptr(), rows, x.shape[-1], out.data_ptr(), _stream_ptr()),
               "nsd_log_softmax")
    return out
