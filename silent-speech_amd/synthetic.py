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


def pack_targets(targets: list[str]):
    """(int8 [N, S] token ids, int32 [N] lengths) of planted target strings."""
    S = max(1, max((len(t) for t in targets), default=1))
    tg = np.zeros((len(targets), S), np.int8)
    tl = np.zeros(len(targets), np.int32)
    for i, t in enumerate(targets):
        ids = [BASE_CHARS.index(c) for c in t]
        tg[i, :len(ids)] = ids
        tl[i] = len(ids)
    return tg, tl
