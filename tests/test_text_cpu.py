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


def test_build_lexicon_filters():
    # parity unpinned: scripts/make_lexicon.py keeps these filters inside main(), which needs packages that
    # are not installed; the expectations below are derived by hand from make_lexicon.py:119-143
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
