"""The CPU restatement (oracle/ns_oracle.py) against the golden vectors that
tests/golden/make_golden.py recorded from the reference decoder itself."""
import numpy as np
import pytest

from oracle import synth
from oracle import ns_oracle as O
from tests.golden_util import case_model, case_targets, decode_kwargs, enc, target_tokens


def _case_ids(golden):
    return [c["name"] for c in golden["cases"]]


def test_every_case_replays_exactly(golden):
    code = golden["meta"]["token_code"]
    n_utts = 0
    for case in golden["cases"]:
        words, trie, lm = case_model(case)
        assert len(words) == case["n_lex_words"]
        targets = case_targets(case, words)
        for u, rec in enumerate(case["utts"]):
            assert targets[u] == rec["target"]
            useed = synth.utt_seed(case["base_seed"], u)
            tgt = target_tokens(rec["target"])
            calls = []

            def step_fn(ids):
                calls.append(enc(ids, code))
                return synth.logits_row(useed, ids, tgt, case["margin"])

            trace = []
            if "raises" in rec:
                with pytest.raises(RuntimeError):
                    O.constrained_beam_decode(step_fn, O.TextTransformLike, case["beam"], case["max_new_tokens"],
                                              trie, lm, trace=trace, **decode_kwargs(case))
                assert calls == rec["calls"], case["name"]
                continue
            ids = O.constrained_beam_decode(step_fn, O.TextTransformLike, case["beam"], case["max_new_tokens"],
                                            trie, lm, trace=trace, **decode_kwargs(case))
            # beam indices: the live hypotheses, in rank order, at every step
            assert calls == rec["calls"], (case["name"], u)
            # trie masks
            masks = [hex(m) for st in trace for m in st["masks"]]
            assert masks == rec["masks"], (case["name"], u)
            assert ids == rec["ids"], (case["name"], u)
            assert O.ids_to_text(O.TextTransformLike, ids) == rec["text"]
            n_utts += 1
    assert n_utts > 80


def test_lm_logp_bit_exact(golden):
    from tests.golden_util import case_words
    for rec in golden["lm_probes"]:
        if rec["n_corpus"]:
            words = synth.make_lexicon(rec["n_words"], rec["lex_seed"], max_len=6)
            corpus = synth.make_sentences(words, rec["n_corpus"], rec["corpus_seed"])
        else:
            corpus = []
        lm = O.NgramLM(rec["order"], floor_vsz=(rec["variant"] == "U")).fit(corpus)
        assert len(lm.V) == rec["vsz"]
        for text, hx in rec["probes"]:
            assert float(lm.logp(text)).hex() == hx, (rec["variant"], rec["order"], text)


def test_postprocess(golden):
    words = synth.make_lexicon(60, 11, max_len=6)
    trie = O.build_trie(words)
    for s, want in golden["postprocess"]:
        assert O.postprocess(s, trie, words) == want


def test_ctc_collapse_known_answers():
    # small readable cases next to the recorded reference outputs of tests/test_oracle_golden_tail.py:
    # repeats are dropped before specials: a,PAD,a -> "aa" but a,a -> "a" (inference_emg_llm.py:447-455)
    TT = O.TextTransformLike

    def frames(tokens):
        x = np.full((len(tokens), TT.VOCAB_SIZE), -5.0, dtype=np.float32)
        for i, t in enumerate(tokens):
            x[i, t] = 3.0
        return x

    assert O.ctc_greedy(frames([0, TT.PAD_IDX, 0]), TT) == [0, 0]
    assert O.ctc_greedy(frames([0, 0]), TT) == [0]
    assert O.ctc_greedy(frames([TT.BOS_IDX, 1, 1, TT.EOS_IDX, 1, TT.PAD_IDX, TT.PAD_IDX, 36, 2]), TT) == [1, 1, 36, 2]
    assert O.ctc_hypothesis(frames([0, TT.PAD_IDX, 0, 36, 36]), TT) == "aa"
    # ties: first maximal index
    x = np.zeros((1, TT.VOCAB_SIZE), dtype=np.float32)
    assert O.ctc_greedy(x, TT) == [0]


def test_rerank_lm_prefers_ar_on_ties():
    lm = O.NgramLM(3).fit(["ab ab", "ab cd"])
    assert O.rerank_lm("ab", "ab", lm) == "ab"
    assert O.rerank_lm("", "ab", lm) == "ab"          # empty string scores -1e9
    assert O.rerank_lm("ab", "", lm) == "ab"
