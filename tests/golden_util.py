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
