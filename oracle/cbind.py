"""TEST INFRASTRUCTURE ONLY — ctypes binding of oracle/ns_oracle_c.c."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "_build" / "libnsoracle.so"
_lib = None


def build(force: bool = False) -> Path:
    src = _DIR / "ns_oracle_c.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_DIR), "-B" if force else "-s"], check=True)
    return _SO


class Params(C.Structure):
    _fields_ = [("variant", C.c_int), ("beam", C.c_int), ("max_new_tokens", C.c_int), ("use_lm", C.c_int),
                ("allow_eos_anytime", C.c_int), ("min_len_for_eos", C.c_int),
                ("beta", C.c_double), ("gamma", C.c_double), ("kappa", C.c_double), ("alpha", C.c_double),
                ("margin", C.c_float), ("lsm_det", C.c_int)]


def lib():
    global _lib
    if _lib is None:
        if not _SO.exists():
            build()
        L = C.CDLL(str(_SO))
        L.nso_log_softmax.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.nso_synth_row.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.nso_hist_next.restype = C.c_uint64
        L.nso_hist_next.argtypes = [C.c_uint64, C.c_int]
        L.nso_utt_seed.restype = C.c_uint64
        L.nso_utt_seed.argtypes = [C.c_uint64, C.c_uint64]
        L.nso_model_create.restype = C.c_void_p
        L.nso_model_create.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.nso_model_destroy.argtypes = [C.c_void_p]
        L.nso_model_nodes.argtypes = [C.c_void_p]
        L.nso_model_vsz.argtypes = [C.c_void_p]
        L.nso_lm_logp.restype = C.c_double
        L.nso_lm_logp.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.nso_decode_synth.argtypes = [C.c_void_p, C.POINTER(Params), C.c_uint64, C.c_int64, C.c_int,
                                       C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.nso_ctc_greedy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def log_softmax(logits):
    x = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().nso_log_softmax(x.ctypes.data, int(x.size), out.ctypes.data)
    return [float(v) for v in out]


def log_softmax_rows(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    V = x.shape[-1]
    flat_in = x.reshape(-1, V)
    flat_out = out.reshape(-1, V)
    for i in range(flat_in.shape[0]):
        L.nso_log_softmax(flat_in[i].ctypes.data, V, flat_out[i].ctypes.data)
    return out


def synth_row(useed: int, hist: int, step: int, tgt_tok: int, margin: float) -> np.ndarray:
    out = np.empty(40, dtype=np.float32)
    lib().nso_synth_row(useed, hist, step, tgt_tok, margin, out.ctypes.data)
    return out


class Model:
    """Trie + n-gram counts held by the C checker (ASCII corpora only)."""

    def __init__(self, words, corpus, order: int, floor_vsz: bool):
        ws = [w.strip().lower() for w in words if (w or "").strip()]
        ts = [(t or "").strip().lower() for t in corpus]
        for t in ts:
            t.encode("ascii")
        wb = b"".join(w.encode("utf-8") + b"\0" for w in ws) + b"\0"
        tb = b"".join(t.encode("ascii") + b"\0" for t in ts) + b"\0"
        self._h = lib().nso_model_create(wb, len(ws), tb, len(ts), int(order), int(bool(floor_vsz)))
        self.order = int(order)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().nso_model_destroy(self._h)
            self._h = None

    def logp(self, text: str) -> float:
        b = text.lower().encode("ascii")
        return float(lib().nso_lm_logp(self._h, b, len(b)))

    @property
    def n_nodes(self):
        return lib().nso_model_nodes(self._h)

    def decode_synth(self, *, variant: str, beam: int, max_new_tokens: int, beta=0.3, gamma=1.0, kappa=0.0,
                     alpha=0.6, min_len_for_eos=8, allow_eos_anytime=True, margin=60.0, base_seed=0, utt0=0,
                     targets=None, target_lens=None, trace=False, threads=None):
        """targets: int8 [N, S] planted token ids, target_lens int32 [N]."""
        targets = np.ascontiguousarray(targets, dtype=np.int8)
        target_lens = np.ascontiguousarray(target_lens, dtype=np.int32)
        N = targets.shape[0]
        T, B = int(max_new_tokens), int(beam)
        P = Params({"U": 0, "I": 1, "E": 2}[variant], B, T, int(self.order > 0), int(allow_eos_anytime),
                   int(min_len_for_eos), beta, gamma, kappa, alpha, margin, 1)
        ids = np.zeros((N, T + 1), dtype=np.uint8)
        lens = np.zeros(N, dtype=np.int32)
        scores = np.zeros(N, dtype=np.float64)
        steps = np.zeros(N, dtype=np.int32)
        tr = {}
        ptrs = [None] * 5
        if trace:
            tr = {"parent": np.full((N, T, B), -2, np.int8), "token": np.full((N, T, B), -2, np.int8),
                  "score": np.zeros((N, T, B), np.float64), "nbeams": np.zeros((N, T), np.int32),
                  "mask": np.zeros((N, T, B), np.uint64)}
            ptrs = [tr[k].ctypes.data for k in ("parent", "token", "score", "nbeams", "mask")]
        rc = lib().nso_decode_synth(self._h, C.byref(P), base_seed, utt0, N, targets.ctypes.data,
                                    targets.shape[1], target_lens.ctypes.data, ids.ctypes.data,
                                    lens.ctypes.data, scores.ctypes.data, steps.ctypes.data, *ptrs)
        if rc != 0:
            raise ValueError(f"nso_decode_synth rejected its arguments ({rc})")
        out = {"ids": ids, "lens": lens, "scores": scores, "steps": steps}
        out.update(tr)
        return out


def ctc_greedy(frame_logits: np.ndarray):
    x = np.ascontiguousarray(frame_logits, dtype=np.float32)
    out = np.zeros(x.shape[0], dtype=np.uint8)
    n = lib().nso_ctc_greedy(x.ctypes.data, x.shape[0], x.shape[1], out.ctypes.data)
    return [int(v) for v in out[:n]]
