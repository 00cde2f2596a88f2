"""The CPU restatement and the host-side helpers of the package against outputs recorded from the reference by
tests/golden/make_golden_tail.py: fp64 scores at benchmark sizes, the CTC proposal and both rerank rules (the
drivers' own statements, inference_emg_llm.py:441-464 and eval_ar.py:554-596), the lexicon filters of
scripts/make_lexicon.py:119-143 and load_lexicon (scripts/ns_utils.py:68-80)."""
import base64
import hashlib

import numpy as np
import pytest

from oracle import synth
from oracle import ns_oracle as O
from tests.golden_util import case_model, case_targets, ctc_frames, decode_kwargs, enc, target_tokens

TT = O.TextTransformLike


@pytest.mark.parametrize("idx", range(21))
def test_scores_and_beam_indices_at_benchmark_sizes(golden_tail, idx):
    """64 utterances of the benchmark workload per copy and beam width: ids, transcript, the fp64 score of the
    returned hypothesis (bit for bit: same torch log-softmax, same operation order), the whole step_fn call
    sequence (digest) and, for two utterances, every kept score of every step."""
    case = golden_tail["scored"][idx]
    code = golden_tail["meta"]["token_code"]
    words, trie, lm = case_model(case)
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
        ids = O.constrained_beam_decode(step_fn, TT, case["beam"], case["max_new_tokens"], trie, lm, trace=trace,
                                        **decode_kwargs(case))
        assert ids == rec["ids"], (case["name"], u)
        assert O.ids_to_text(TT, ids) == rec["text"]
        assert len(trace) == rec["steps"]
        assert float(trace[-1]["chosen"][0][2]).hex() == rec["score"], (case["name"], u)
        assert len(calls) == rec["n_calls"]
        assert hashlib.sha1("\n".join(calls).encode()).hexdigest() == rec["calls_sha1"], (case["name"], u)
        if "kept_scores_f64" in rec:
            want = np.frombuffer(base64.b64decode(rec["kept_scores_f64"]), np.float64)
            got = np.array([c[2] for st in trace for c in st["chosen"]], np.float64)
            assert [len(st["chosen"]) for st in trace] == rec["kept_counts"]
            assert got.tobytes() == want.tobytes(), (case["name"], u)


def _tail_model(doc):
    words = synth.make_lexicon(doc["n_words"], doc["lex_seed"], max_len=doc["max_len"])
    corpus = synth.make_sentences(words, doc["n_corpus"], doc["corpus_seed"])
    return words, corpus


@pytest.mark.parametrize("which", ["I", "E"])
def test_ctc_proposal_and_rerank_equal_the_drivers_statements(golden_tail, which):
    doc = golden_tail["ctc_tail_" + which]
    words, corpus = _tail_model(doc)
    trie = O.build_trie(words)
    lm = O.NgramLM(doc["order"], floor_vsz=False).fit(corpus)
    assert synth.make_sentences(words, doc["n_utts"], doc["target_seed"], 1, 4) == [r["target"] for r in doc["rows"]]
    n_ctc = 0
    for u, r in enumerate(doc["rows"]):
        x = ctc_frames(r["ctc_seed"], r["ctc_target"], r["Tp"], wrong=r["wrong"])
        assert O.ctc_greedy(x, TT) == r["ctc_ids"], u
        hyp_ctc = O.postprocess(O.ctc_hypothesis(x, TT), trie, words)
        assert hyp_ctc == r["hyp_ctc"], u
        assert O.postprocess(r["hyp_ar_raw"], trie, words) == r["hyp_ar"]
        if which == "I":
            assert O.rerank_lm(r["hyp_ar"], hyp_ctc, lm) == r["hyp"], u
            sc = [float(lm.logp(s)) - 0.002 * len(s) for s in (r["hyp_ar"], hyp_ctc)]
        else:
            useed = synth.utt_seed(doc["step_seed"], u)
            tgt = target_tokens(r["target"])
            step_fn = lambda ids: synth.logits_row(useed, ids, tgt, doc["step_margin"])      # noqa: E731
            assert O.rerank_joint(r["hyp_ar"], hyp_ctc, step_fn, TT, lm, doc["beta"]) == r["hyp"], u
            sc = [O.ar_logprob(s, step_fn, TT) + doc["beta"] * float(lm.logp(s)) for s in (r["hyp_ar"], hyp_ctc)]
        assert [float(v).hex() for v in sc] == r["scores"], u
        n_ctc += int(r["hyp"] != r["hyp_ar"])
    assert 0 < n_ctc < len(doc["rows"])


def test_driver_tail_of_the_package_equals_the_reference(golden_tail):
    """text.finish_transcripts (ids_to_text + postprocess + the LM choice) on the recorded inputs of the
    inference driver's statements."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd import tables as T
    from silent_speech_amd.text import finish_transcripts
    doc = golden_tail["ctc_tail_I"]
    words, corpus = _tail_model(doc)
    trie = O.build_trie(words)
    nt = T.compile_ngram(T.NgramCounts(doc["order"], floor_vsz=False).fit(corpus))
    rows = doc["rows"]
    ar = [[37] + [synth.BASE_CHARS.index(c) for c in r["hyp_ar_raw"]] + [38] for r in rows]
    ctc = [r["ctc_ids"] for r in rows]
    got = finish_transcripts(TextTransform, ar, [len(a) for a in ar], trie, words, lambda s: T.host_logp(nt, s),
                             ctc, [len(c) for c in ctc])
    assert got == [r["hyp"] for r in rows]


def test_build_lexicon_equals_make_lexicon(golden_tail):
    from silent_speech_amd.text import build_lexicon
    assert len(golden_tail["make_lexicon"]) >= 7
    for rec in golden_tail["make_lexicon"]:
        got = build_lexicon(rec["texts"], min_freq=rec["min_freq"], min_len=rec["min_len"],
                            max_words=rec["max_words"], add_words=rec["add_words"])
        assert got == rec["lex_words"], rec["name"]


def test_load_lexicon_words_equals_load_lexicon(golden_tail, tmp_path):
    from silent_speech_amd import tables as T
    for k, rec in enumerate(golden_tail["load_lexicon"]):
        lex = tmp_path / f"lex{k}.txt"
        lex.write_text("\n".join(rec["lexicon"]) + "\n", encoding="utf-8")
        extra = None
        if isinstance(rec["phrasebook"], list):
            extra = tmp_path / f"pb{k}.txt"
            extra.write_text("\n".join(rec["phrasebook"]) + "\n", encoding="utf-8")
        elif rec["phrasebook"] == "missing":
            extra = tmp_path / "does_not_exist.txt"
        words = T.load_lexicon_words(str(lex), str(extra) if extra else None)
        assert words == rec["words"]
        # the flat trie built from those words answers like the reference's Trie
        trie = O.build_trie(words)
        for p, want in rec["is_word"]:
            assert bool(trie.is_word(p)) == want, p
        for p, want in rec["next_chars"]:
            assert "".join(sorted(trie.next_chars(p))) == want, p
        nodes = T.flatten_trie(T.HostTrie.from_words(words))
        for p, want in rec["is_word"]:
            if p and all(c in synth.BASE_CHARS[:36] for c in p):
                n = T.host_trie_walk(nodes, p)
                assert (n >= 0 and bool((int(nodes["mask"][n]) >> 37) & 1)) == want, p
