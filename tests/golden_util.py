"""Helpers shared by the tests that replay the golden cases."""
from __future__ import annotations

import numpy as np

from oracle import synth
from oracle.ns_oracle import NgramLM, build_trie


def case_words(case):
    words = synth.make_lexicon(case["n_words"], case["lex_seed"],
                               case.get("alphabet", synth.BASE_CHARS[:26]),
                               case.get("min_len", 2), case.get("max_len", 6))
    return sorted(set(words + case.get("extra_words", [])))


def case_corpus(case, words):
    plain = [w for w in words if all(c in synth.BASE_CHARS for c in w)]
    return synth.make_sentences(plain, case["n_corpus"], case["corpus_seed"]) if plain else []


def case_model(case):
    """(words, trie, lm) rebuilt with the oracle's own classes."""
    words = case_words(case)
    trie = build_trie(words)
    lm = None
    if case["lm_order"] > 0:
        lm = NgramLM(case["lm_order"], floor_vsz=(case["variant"] == "U")).fit(case_corpus(case, words))
    return words, trie, lm


def case_targets(case, words):
    """Planted target strings, as make_golden.py chose them."""
    if case.get("targets"):
        return list(case["targets"])
    plain = [w for w in words if all(c in synth.BASE_CHARS for c in w)]
    return synth.make_sentences(plain, case["n_utts"], case["target_seed"],
                                case.get("min_words", 2), case.get("max_words", 4))


def target_tokens(tgt: str):
    return [synth.BASE_CHARS.index(c) for c in tgt if c in synth.BASE_CHARS]


def decode_kwargs(case):
    kw = dict(lm_weight=case["beta"], oov_penalty=case["gamma"], variant=case["variant"])
    if case["variant"] == "E":
        kw.update(word_bonus=case["kappa"], length_penalty_alpha=case["alpha"],
                  min_len_for_eos=case["min_len_for_eos"])
    return kw


def enc(ids, code):
    return "".join(code[i] for i in ids[1:])


def ctc_frames(seed: int, target: str, Tp: int, noise: float = 0.5, wrong: float = 0.05) -> np.ndarray:
    """Seeded (Tp, 40) float32 CTC logits with a planted frame path: every target character repeated over 1-3
    frames, blanks (PAD, id 39) between equal neighbours and here and there, a few BOS/EOS frames and a few
    wrong characters; the rest of the frames are blanks.  numpy's legacy RandomState keeps this stable."""
    rs = np.random.RandomState(seed)
    path = []
    prev = None
    for tok in target_tokens(target):
        if tok == prev or rs.rand() < 0.3:
            path += [synth.PAD_ID] * int(rs.randint(1, 3))
        if rs.rand() < wrong:
            tok_f = int(rs.randint(0, 37))
        else:
            tok_f = tok
        path += [tok_f] * int(rs.randint(1, 4))
        if rs.rand() < 0.05:
            path.append(synth.BOS_ID if rs.rand() < 0.5 else synth.EOS_ID)
        prev = tok
    path = (path + [synth.PAD_ID] * Tp)[:Tp]
    x = (rs.standard_normal((Tp, synth.VOCAB)) * noise).astype(np.float32)
    x[np.arange(Tp), np.array(path)] += np.float32(6.0)
    return x


def corrupt(seed: int, text: str, rate: float = 0.12) -> str:
    """A transcript with a few wrong, doubled or tripled letters (what an AR hypothesis looks like before the
    post-correction)."""
    rs = np.random.RandomState(seed)
    out = []
    for ch in text:
        r = rs.rand()
        if r < rate / 3:
            out.append(synth.BASE_CHARS[int(rs.randint(0, 26))])
        elif r < 2 * rate / 3:
            out += [ch, ch, ch]
        elif r < rate:
            continue
        else:
            out.append(ch)
    return "".join(out)
