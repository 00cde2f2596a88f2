"""Host-side text helpers around the decoder: post-correction and lexicon building."""
from __future__ import annotations

import difflib
from collections import Counter
from typing import Iterable, List, Optional, Sequence

from .tables import BASE_CHARS


def postprocess(text: str, is_word, lex_words: Sequence[str]) -> str:
    """Deterministic clean-up of a decoded transcript (scripts/ns_utils.py:166-188).

    Whitespace runs collapse to one space; a letter repeated three or more times in a row keeps two;
    a word outside the lexicon is replaced by its closest lexicon word when difflib's similarity ratio
    reaches 0.82.  ``is_word`` is a callable (e.g. ``trie.is_word``) or any object with ``.is_word``.
    """
    check = is_word.is_word if hasattr(is_word, "is_word") else is_word
    out: List[str] = []
    for w in " ".join((text or "").split()).split():
        kept: List[str] = []
        for ch in w:
            if len(kept) >= 2 and kept[-1] == ch and kept[-2] == ch:
                continue
            kept.append(ch)
        w2 = "".join(kept)
        if not check(w2):
            near = difflib.get_close_matches(w2, lex_words, n=1, cutoff=0.82)
            if near:
                w2 = near[0]
        out.append(w2)
    return " ".join(out)


def build_lexicon(texts: Iterable[str], min_freq: int = 1, min_len: int = 1, max_words: int = 200000,
                  add_words: Optional[Iterable[str]] = None) -> List[str]:
    """Word list from already-cleaned transcripts, with the filters of scripts/make_lexicon.py:119-143:
    count whitespace-separated words of at least ``min_len`` characters, keep those seen ``min_freq`` times
    whose characters all belong to the decoder's alphabet, order by (-count, word), cap at ``max_words``,
    then append extra words (same alphabet filter) without duplicates."""
    cnt: Counter = Counter()
    for t in texts:
        for w in t.split():
            if len(w) >= min_len:
                cnt[w] += 1
    allowed = set(BASE_CHARS)
    words = [w for w, c in cnt.items() if c >= min_freq and all(ch in allowed for ch in w)]
    words.sort(key=lambda w: (-cnt[w], w))
    if max_words and max_words > 0:
        words = words[:max_words]
    if add_words is not None:
        extras = [l.strip().lower() for l in add_words if l.strip()]
        extras = [w for w in extras if all(ch in allowed for ch in w)]
        words = list(dict.fromkeys(words + extras))
    return words
