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


def ids_to_text(TT, ids: Sequence[int]) -> str:
    """scripts/ns_utils.py:284-296: skip ids[0], stop at EOS, keep base characters, strip."""
    out = []
    for i in ids[1:]:
        if i == TT.EOS_IDX:
            break
        j = i - TT.VOCAB_OFFSET
        if 0 <= j < len(TT.BASE_CHARS):
            out.append(TT.BASE_CHARS[j])
    return "".join(out).strip()


def rerank_by_lm(hyp_ar: str, hyp_ctc: str, logp) -> str:
    """The driver's choice between the AR and the CTC transcript (inference_emg_llm.py:457-464):
    ``logp(s) - 0.002 * len(s)``, the AR transcript wins ties (first maximal element)."""
    def score(s: str) -> float:
        return float(logp(s)) - 0.002 * len(s)
    return max([(hyp_ar, score(hyp_ar)), (hyp_ctc, score(hyp_ctc))], key=lambda c: c[1])[0]


def finish_transcripts(TT, ar_ids, ar_len, is_word, lex_words: Sequence[str], logp=None,
                       ctc_ids=None, ctc_len=None) -> List[str]:
    """What the driver does with one utterance after the beam search (inference_emg_llm.py:436-464), for a
    batch: ``ids_to_text`` + ``postprocess`` of the AR ids and, when collapsed CTC ids are given
    (``--use_ctc_rerank``), the same for the CTC proposal wrapped in BOS/EOS and the LM choice between the
    two.  ``ar_ids`` [N, >=len] starts with BOS; ``ctc_ids`` [N, >=len] holds the ids ``ctc_greedy_batch``
    returns (repeats collapsed, PAD/BOS/EOS dropped).  ``logp`` is ``lm.logp`` (any callable str -> float).
    Rows may be numpy arrays, tensors moved to the host, or lists."""
    out: List[str] = []
    for u in range(len(ar_len)):
        hyp = postprocess(ids_to_text(TT, [int(t) for t in ar_ids[u][:int(ar_len[u])]]), is_word, lex_words)
        if ctc_ids is not None:
            if logp is None:
                raise ValueError("the CTC rerank needs the LM's logp")
            ids = [TT.BOS_IDX] + [int(t) for t in ctc_ids[u][:int(ctc_len[u])]] + [TT.EOS_IDX]
            hyp_ctc = postprocess(ids_to_text(TT, ids), is_word, lex_words)
            hyp = rerank_by_lm(hyp, hyp_ctc, logp)
        out.append(hyp)
    return out
