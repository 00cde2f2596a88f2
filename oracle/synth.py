"""TEST INFRASTRUCTURE ONLY — seeded synthetic workload for the NS decoder.

Everything here is integer-hash based so that the CPU checker and the HIP
logits kernel produce bit-identical fp32 logits with no transcendental calls:

  * lexicon   : random words over a-z (closed template) or a-z0-9 (open)
  * sentences : 3-7 lexicon words joined by single spaces
  * logits    : logit[v] = 4*(r/2^24 - 0.5) (+ margin if v is the planted
                target token of this step), r = top 24 bits of a 32-bit hash
                of (utterance seed, step, token-history hash, v)

The logits of a beam depend only on its token history `ids`, so the same
generator serves the reference-style per-beam `step_fn(ids)` and the batched
device kernel (which carries the rolling history hash in the beam state).

The vocabulary follows the reference's TextTransform
(/root/reference/scripts/data_utils.py:44-50): ids 0-25 a-z, 26-35 0-9,
36 space, 37 BOS, 38 EOS, 39 PAD.
"""
from __future__ import annotations

import numpy as np

BASE_CHARS = "abcdefghijklmnopqrstuvwxyz0123456789 "
SPACE_ID = 36
BOS_ID = 37
EOS_ID = 38
PAD_ID = 39
VOCAB = 40

_M64 = (1 << 64) - 1
_GOLD = 0x9E3779B97F4A7C15
HIST0 = 0x6A09E667F3BCC908  # history hash of the empty history [BOS]


def mix64(z: int) -> int:
    """splitmix64 finaliser on Python ints (mod 2^64)."""
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def fmix32(h: int) -> int:
    """murmur3 32-bit finaliser."""
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    return h ^ (h >> 16)


def hist_next(h: int, tok: int) -> int:
    """Rolling hash of the generated-token history (BOS excluded)."""
    return mix64(h ^ (((tok + 1) * _GOLD) & _M64))


def hist_of(ids) -> int:
    h = HIST0
    for t in ids[1:]:
        h = hist_next(h, int(t))
    return h


def utt_seed(base_seed: int, utt: int) -> int:
    return mix64((base_seed + utt * 0xD1B54A32D192ED03) & _M64)


def row_key(useed: int, step: int, hist: int) -> int:
    """32-bit key of one logits row."""
    k = mix64(hist ^ mix64((useed + step * _GOLD) & _M64))
    return (k ^ (k >> 32)) & 0xFFFFFFFF


def logits_row(useed: int, ids, target, margin: float) -> np.ndarray:
    """fp32 logits (VOCAB,) for the beam whose token history is `ids`.

    `target` is the planted token sequence (no BOS/EOS); past its end the
    planted token is EOS.
    """
    step = len(ids) - 1
    rk = row_key(useed, step, hist_of(ids))
    v = np.arange(VOCAB, dtype=np.uint64)
    h = (np.uint64(rk) + v * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h = _fmix32_np(h)
    r = (h >> np.uint64(8)).astype(np.float32)           # < 2^24: exact
    x = (r * np.float32(2.0 ** -24) - np.float32(0.5)) * np.float32(4.0)
    tgt = int(target[step]) if step < len(target) else EOS_ID
    x[tgt] = np.float32(x[tgt] + np.float32(margin))
    return x.astype(np.float32)


def _fmix32_np(h):
    m = np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & m
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & m
    return h ^ (h >> np.uint64(16))


# ----------------------------------------------------------------------
# lexicon / sentences / corpus
# ----------------------------------------------------------------------
def make_lexicon(n_words: int, seed: int, alphabet: str = "abcdefghijklmnopqrstuvwxyz",
                 min_len: int = 2, max_len: int = 8) -> list[str]:
    """`n_words` distinct random words, sorted (as load_lexicon sorts them)."""
    rng = np.random.RandomState(seed)
    words: set[str] = set()
    A = len(alphabet)
    while len(words) < n_words:
        need = n_words - len(words)
        lens = rng.randint(min_len, max_len + 1, size=need + 16)
        for L in lens:
            idx = rng.randint(0, A, size=int(L))
            words.add("".join(alphabet[i] for i in idx))
            if len(words) >= n_words:
                break
    return sorted(words)


def make_sentences(words: list[str], n: int, seed: int,
                   min_words: int = 3, max_words: int = 7) -> list[str]:
    rng = np.random.RandomState(seed)
    out = []
    W = len(words)
    for _ in range(n):
        k = int(rng.randint(min_words, max_words + 1))
        out.append(" ".join(words[int(i)] for i in rng.randint(0, W, size=k)))
    return out


def text_to_tokens(s: str) -> list[int]:
    return [BASE_CHARS.index(c) for c in s]


def ids_to_text(ids) -> str:
    """Restates ids_to_text, /root/reference/scripts/ns_utils.py:284-296."""
    out = []
    for i in ids[1:]:
        if i == EOS_ID:
            break
        if 0 <= i < len(BASE_CHARS):
            out.append(BASE_CHARS[i])
    return "".join(out).strip()
