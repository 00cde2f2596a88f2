"""Host-side text helpers and the table file format (no GPU needed)."""
import numpy as np

from oracle import synth
from oracle import ns_oracle as O
from silent_speech_amd import tables as T
from silent_speech_amd.text import build_lexicon, postprocess


def test_postprocess_matches_reference_outputs(golden):
    # outputs recorded from the reference's own postprocess (scripts/ns_utils.py:166-188)
    words = synth.make_lexicon(60, 11, max_len=6)
    trie = O.build_trie(words)
    assert len(golden["postprocess"]) >= 15
    for s, want in golden["postprocess"]:
        assert postprocess(s, trie, words) == want
        assert postprocess(s, trie.is_word, words) == want


def test_build_lexicon_small_cases():
    # readable cases; the outputs recorded from scripts/make_lexicon.py:119-143 itself are replayed in
    # tests/test_oracle_golden_tail.py::test_build_lexicon_equals_make_lexicon
    texts = ["the cat sat", "the cat", "a b the", "naïve the don't", "zz zz zz", "the"]
    assert build_lexicon(texts) == ["the", "zz", "cat", "a", "b", "sat"]
    assert build_lexicon(texts, min_freq=2) == ["the", "zz", "cat"]
    assert build_lexicon(texts, min_len=2, max_words=2) == ["the", "zz"]
    assert build_lexicon(texts, min_freq=3, add_words=[" Dog ", "cat", "x-ray", "", "the"]) == ["the", "zz", "dog", "cat"]
    assert build_lexicon([]) == []


def test_table_file_round_trip(tmp_path):
    words = synth.make_lexicon(300, 4) + ["new york", "don't"]
    lm = O.NgramLM(5).fit(synth.make_sentences(words[:300], 500, 2))
    nodes = T.flatten_trie(T.HostTrie.from_words(words))
    ng = T.compile_ngram(lm)
    p = tmp_path / "tables.npz"
    T.save_tables(str(p), nodes, ng)
    n2, g2 = T.load_tables(str(p))
    assert n2.tobytes() == nodes.tobytes()
    for name in ("grams", "ctxs", "ctx_recs", "gram_recs"):
        assert getattr(g2, name).tobytes() == getattr(ng, name).tobytes()
    assert (g2.order, g2.vsz, g2.unseen, g2.init_ctx) == (ng.order, ng.vsz, ng.unseen, ng.init_ctx)
    T.save_tables(str(p), nodes, None)
    n3, g3 = T.load_tables(str(p))
    assert g3 is None and n3.tobytes() == nodes.tobytes()


def test_finish_transcripts_equals_the_driver_tail():
    """ids_to_text + postprocess of the AR ids, the same for the wrapped CTC ids, then the LM choice
    (inference_emg_llm.py:436-464), against the restatement of each piece; the LM scores of the flat tables equal
    CharNgramLM.logp, so the choice is the same with either."""
    from silent_speech_amd import TextTransform as TT
    from silent_speech_amd.text import finish_transcripts, rerank_by_lm
    rng = np.random.RandomState(3)
    words = synth.make_lexicon(80, 5, max_len=6)
    corpus = synth.make_sentences(words, 500, 2)
    trie = O.build_trie(words)
    lm = O.NgramLM(3, floor_vsz=False).fit(corpus)
    nt = T.compile_ngram(lm, floor_vsz=False)
    N, L = 60, 30
    targets = synth.make_sentences(words, N, 9, 1, 4)
    ar = np.full((N, L + 2), TT.PAD_IDX, np.uint8)
    ar_len = np.zeros(N, np.int32)
    ctc = np.zeros((N, L), np.uint8)
    ctc_len = np.zeros(N, np.int32)
    for u, t in enumerate(targets):
        toks = synth.text_to_tokens(t)[:L]
        noisy = [int(rng.randint(0, 37)) if rng.rand() < 0.15 else k for k in toks]      # a few wrong letters
        if u % 7 == 0:
            noisy = noisy[:1] + noisy[:1] * 3 + noisy[1:]                                 # a run of four equal letters
        seq = [TT.BOS_IDX] + noisy[:L] + [TT.EOS_IDX]
        ar[u, :len(seq)] = seq
        ar_len[u] = len(seq) if u % 5 else len(seq) - 1                                   # some without EOS
        c = toks if u % 3 else [int(rng.randint(0, 37)) for _ in range(rng.randint(0, 6))]
        if u % 11 == 0:
            c = []
        ctc[u, :len(c)] = c
        ctc_len[u] = len(c)
    got = finish_transcripts(TT, ar, ar_len, trie, words, lm.logp, ctc, ctc_len)
    got_tables = finish_transcripts(TT, ar, ar_len, trie.is_word, words, lambda s: T.host_logp(nt, s), ctc, ctc_len)
    got_ar_only = finish_transcripts(TT, ar, ar_len, trie, words)
    n_ctc = 0
    for u in range(N):
        hyp_ar = O.postprocess(O.ids_to_text(O.TextTransformLike, ar[u, :ar_len[u]].tolist()), trie, words)
        wrapped = [TT.BOS_IDX] + ctc[u, :ctc_len[u]].tolist() + [TT.EOS_IDX]
        hyp_ctc = O.postprocess(O.ids_to_text(O.TextTransformLike, wrapped), trie, words)
        want = O.rerank_lm(hyp_ar, hyp_ctc, lm)
        assert got[u] == want and got_tables[u] == want, u
        assert got_ar_only[u] == hyp_ar
        n_ctc += int(want != hyp_ar)
    assert 0 < n_ctc < N
    assert rerank_by_lm("ab", "ab", lm.logp) == "ab"
