"""Seeded synthetic workload for benchmarks: lexicon, sentences, planted targets.

The logits themselves come from the device kernel nsd_synth_logits (integer
hash of utterance seed, step and token history, plus a margin on the planted
target token); this module only makes the text side with numpy's MT19937
RandomState, which is stable across numpy versions.
"""
from __future__ import annotations

import numpy as np

from .tables import BASE_CHARS


def make_lexicon(n_words: int, seed: int, alphabet: str = BASE_CHARS[:26],
                 min_len: int = 2, max_len: int = 8) -> list[str]:
    """`n_words` distinct random words, sorted."""
    rng = np.random.RandomState(seed)
    words: set[str] = set()
    A = len(alphabet)
    while len(words) < n_words:
        need = n_words - len(words)
        for L in rng.randint(min_len, max_len + 1, size=need + 16):
            idx = rng.randint(0, A, size=int(L))
            words.add("".join(alphabet[i] for i in idx))
            if len(words) >= n_words:
                break
    return sorted(words)


def make_sentences(words: list[str], n: int, seed: int, min_words: int = 3, max_words: int = 7) -> list[str]:
    rng = np.random.RandomState(seed)
    W = len(words)
    out = []
    for _ in range(n):
        k = int(rng.randint(min_words, max_words + 1))
        out.append(" ".join(words[int(i)] for i in rng.randint(0, W, size=k)))
    return out


TARGET_BLOCK = 4096


def targets_for_range(words: list[str], lo: int, hi: int, min_words: int = 3, max_words: int = 7) -> list[str]:
    """Planted target sentences of the utterances [lo, hi) of the global numbering.

    Utterances are drawn in blocks of TARGET_BLOCK with one seed per block (1000 + block number), so the
    target of utterance u does not depend on how the utterances are split over ranks."""
    out: list[str] = []
    if hi <= lo:
        return out
    for k in range(lo // TARGET_BLOCK, (hi - 1) // TARGET_BLOCK + 1):
        b0 = k * TARGET_BLOCK
        block = make_sentences(words, TARGET_BLOCK, 1000 + k, min_words, max_words)
        out.extend(block[max(lo, b0) - b0:min(hi, b0 + TARGET_BLOCK) - b0])
    return out


_TOKEN_OF_BYTE = np.full(256, -1, np.int16)
for _i, _c in enumerate(BASE_CHARS):
    _TOKEN_OF_BYTE[ord(_c)] = _i


def pack_targets(targets: list[str]):
    """(int8 [N, S] token ids, int32 [N] lengths) of planted target strings."""
    n = len(targets)
    tl = np.fromiter((len(t) for t in targets), np.int32, count=n)
    S = max(1, int(tl.max()) if n else 1)
    tg = np.zeros((n, S), np.int8)
    if n and int(tl.sum()):
        flat = _TOKEN_OF_BYTE[np.frombuffer("".join(targets).encode("ascii"), np.uint8)]
        if (flat < 0).any():
            raise ValueError("target outside the base characters")
        rows = np.repeat(np.arange(n), tl)
        starts = np.cumsum(tl) - tl
        cols = np.arange(flat.shape[0]) - np.repeat(starts, tl)
        tg[rows, cols] = flat.astype(np.int8)
    return tg, tl


def ctc_frame_logits(tg: np.ndarray, tl: np.ndarray, utt0: int, frames: int = 256, seed: int = 7,
                     margin: float = 3.98, sub_rate: float = 0.02) -> np.ndarray:
    """Synthetic CTC-head output, float32 [N, frames, 40], for planted targets (int8 [N, S], int32 [N]).

    Each target character is held for 1-3 frames, a blank (id 39, the CTC blank of train_emg_llm.py:239) separates
    equal neighbours, 0-2 blanks follow every character, blanks fill the rest; with probability `sub_rate` a
    character is replaced by a random one, so some proposals carry out-of-lexicon words.  Logits are uniform noise in
    [-2, 2) plus `margin` on the frame's label: with a margin just below 4 a frame in a few hundred is won by
    the noise.
    One MT19937 stream per utterance of the global numbering (seed, utt0 + u), independent of the batch split."""
    n = int(tl.shape[0])
    out = np.empty((n, frames, 40), np.float32)
    blank = 39
    rows = np.arange(frames)
    for u in range(n):
        rng = np.random.RandomState((seed * 1000003 + utt0 + u) & 0x7FFFFFFF)
        labels: list[int] = []
        for c in tg[u, :int(tl[u])].tolist():
            if rng.random_sample() < sub_rate:
                c = int(rng.randint(0, 37))
            if labels and labels[-1] == c:
                labels.append(blank)
            labels.extend([c] * int(rng.randint(1, 4)))
            labels.extend([blank] * int(rng.randint(0, 3)))
        lab = np.full(frames, blank, np.int64)
        k = min(frames, len(labels))
        lab[:k] = labels[:k]
        x = ((rng.random_sample((frames, 40)) - 0.5) * 4.0).astype(np.float32)
        x[rows, lab] += np.float32(margin)
        out[u] = x
    return out
