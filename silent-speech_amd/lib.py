"""ctypes binding of libnsdecode.so (see include/nsdecode.h).

There is no CPU fall-back: if the shared library is missing, loading fails
with an error that says how to build it.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
# NSDECODE_LIB selects another build of the same library (diagnostic and experimental builds)
SO_PATH = Path(os.environ.get("NSDECODE_LIB") or PKG / "libnsdecode.so")

MAX_BEAM = 64
MAX_STEPS = 128
TOKEN_CARRY = 0xFF
U_DONE, U_ERR_TOPK, U_ERR_INTERNAL, U_STEPS_SHIFT = 0x100, 0x200, 0x400, 16
F_FINISHED, F_DEAD, F_PENDING_SPACE = 1, 2, 4
E_ALIGN = -9
VARIANTS = {"U": 0, "I": 1, "E": 2}
PRIOR_FORMS = {"U": 0, "T": 1}


class NsdError(RuntimeError):
    pass


class TrieCSR(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint32)]


class NgramHash(C.Structure):
    _fields_ = [("order", C.c_int32), ("init_ctx", C.c_uint32), ("gram_cap", C.c_uint64), ("ctx_cap", C.c_uint64),
                ("grams", C.c_void_p), ("ctxs", C.c_void_p), ("unseen", C.c_double),
                ("n_ctx_recs", C.c_uint32), ("n_gram_recs", C.c_uint32), ("ctx_recs", C.c_void_p),
                ("gram_recs", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("variant", C.c_int32), ("use_lm", C.c_int32), ("allow_eos_anytime", C.c_int32),
                ("min_len_for_eos", C.c_int32), ("lanes_per_utt", C.c_int32), ("lds_trie_nodes", C.c_int32),
                ("beta", C.c_double), ("gamma", C.c_double),
                ("kappa", C.c_double), ("eos_len_nudge", C.c_double), ("len_denom", C.c_double * (MAX_STEPS + 4))]


class BeamStatePtrs(C.Structure):
    _fields_ = [("score", C.c_void_p), ("lmsum", C.c_void_p), ("lmsum_e", C.c_void_p), ("ctx", C.c_void_p),
                ("hist", C.c_void_p), ("ctx_id", C.c_void_p), ("ctxe_id", C.c_void_p), ("node", C.c_void_p),
                ("meta", C.c_void_p), ("utt", C.c_void_p), ("done_count", C.c_void_p)]


EXPORTS = ["nsd_version", "nsd_strerror", "nsd_hash64", "nsd_tables_create", "nsd_tables_destroy",
           "nsd_tables_order", "nsd_state_init", "nsd_step", "nsd_backtrace", "nsd_gather_ids",
           "nsd_ctc_greedy", "nsd_lm_logp", "nsd_rerank_gather", "nsd_log_softmax", "nsd_synth_logits",
           "nsd_decode_synth", "nsd_rerank_lm", "nsd_cond_dist", "nsd_lm_prior_ce", "nsd_lm_prior_ce_grad", "nsd_kv_reorder",
           "nsd_shutdown", "nsd_decode_synth_noalloc", "nsd_ancestors", "nsd_add_rmsnorm", "nsd_rope_kv_append",
           "nsd_attn_step", "nsd_attn_ctx", "nsd_swiglu"]

_lib = None


def load():
    """The loaded library; raises NsdError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not SO_PATH.exists():
        raise NsdError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fall-back for the NS decoder")
    L = C.CDLL(str(SO_PATH))
    vp, i32, u64, i64 = C.c_void_p, C.c_int32, C.c_uint64, C.c_int64
    L.nsd_version.restype = C.c_int
    L.nsd_strerror.restype = C.c_char_p
    L.nsd_strerror.argtypes = [C.c_int]
    L.nsd_hash64.restype = u64
    L.nsd_hash64.argtypes = [u64]
    L.nsd_tables_create.argtypes = [C.POINTER(TrieCSR), C.POINTER(NgramHash), C.POINTER(vp)]
    L.nsd_tables_destroy.argtypes = [vp]
    L.nsd_tables_order.argtypes = [vp]
    L.nsd_state_init.argtypes = [vp, C.POINTER(BeamStatePtrs), i32, i32, vp]
    L.nsd_step.argtypes = [vp, C.POINTER(Params), vp, C.POINTER(BeamStatePtrs), C.POINTER(BeamStatePtrs),
                           vp, vp, vp, i32, i32, i32, vp]
    L.nsd_backtrace.argtypes = [C.POINTER(BeamStatePtrs), vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.nsd_gather_ids.argtypes = [C.POINTER(BeamStatePtrs), vp, vp, i32, i32, i32, vp, vp, vp]
    L.nsd_ctc_greedy.argtypes = [vp, i32, i32, i32, vp, vp, vp]
    L.nsd_lm_logp.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    L.nsd_rerank_gather.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.nsd_log_softmax.argtypes = [vp, i32, i32, vp, vp]
    L.nsd_synth_logits.argtypes = [C.POINTER(BeamStatePtrs), u64, i64, vp, vp, i32, C.c_float, vp, i32, i32, vp]
    L.nsd_decode_synth.argtypes = [vp, C.POINTER(Params), C.POINTER(BeamStatePtrs), u64, i64, vp, vp, i32, C.c_float,
                                   vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp, vp, vp]
    L.nsd_decode_synth_noalloc.argtypes = [vp, C.POINTER(Params), u64, i64, vp, vp, i32, C.c_float, i32, i32, i32,
                                           vp, vp, vp, vp]
    L.nsd_ancestors.argtypes = [vp, i32, i32, i32, vp, vp]
    L.nsd_add_rmsnorm.argtypes = [vp, vp, vp, C.c_float, i32, i32, i32, vp, vp, vp]
    L.nsd_rope_kv_append.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    L.nsd_attn_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_float, i32, vp, vp, vp,
                                vp]
    L.nsd_attn_ctx.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp, vp, vp, vp]
    L.nsd_swiglu.argtypes = [vp, i64, i32, i32, vp, vp]
    L.nsd_rerank_lm.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp]
    L.nsd_kv_reorder.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.nsd_cond_dist.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.nsd_lm_prior_ce.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.nsd_lm_prior_ce_grad.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().nsd_strerror(code).decode()
        raise NsdError(f"{what} failed: {msg} ({code})")


def make_params(variant: str, *, use_lm: bool, beta: float, gamma: float, kappa: float = 0.0,
                alpha: float = 0.6, min_len_for_eos: int = 8, allow_eos_anytime: bool = True,
                lanes_per_utt: int = 0, lds_trie_nodes: int = 0) -> Params:
    p = Params()
    p.lanes_per_utt = int(lanes_per_utt)
    p.lds_trie_nodes = int(lds_trie_nodes)
    p.variant = VARIANTS[variant]
    p.use_lm = int(bool(use_lm))
    p.allow_eos_anytime = int(bool(allow_eos_anytime))
    p.min_len_for_eos = int(min_len_for_eos)
    p.beta, p.gamma, p.kappa = float(beta), float(gamma), float(kappa)
    p.eos_len_nudge = 0.002
    # the reference's normaliser, evaluated with the reference's own expression
    # (eval_ar.py:237-241) so that the device divides by the very same doubles
    for L in range(1, MAX_STEPS + 4):
        p.len_denom[L] = ((5 + L) ** alpha) / ((5 + 1) ** alpha)
    return p


def np_ptr(a: np.ndarray) -> int:
    return a.ctypes.data
