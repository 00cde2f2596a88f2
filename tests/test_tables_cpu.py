"""Table compiler (host) and C-ABI surface; no GPU needed."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import synth
from oracle import ns_oracle as O
from silent_speech_amd import tables as T

ROOT = Path(__file__).resolve().parent.parent


def test_flat_trie_matches_dict_trie():
    words = synth.make_lexicon(400, 5, max_len=7) + ["don't", "new york", "a b", "été", "x"]
    trie = O.build_trie(words)
    nodes = T.flatten_trie(trie)
    assert nodes.dtype.itemsize == 16
    prefixes = {""}
    for w in words:
        for i in range(1, len(w) + 1):
            prefixes.add(w[:i])
    rng = np.random.RandomState(1)
    for _ in range(300):
        prefixes.add("".join(synth.BASE_CHARS[i] for i in rng.randint(0, 36, rng.randint(1, 4))))
    for p in sorted(prefixes):
        if any(c not in synth.BASE_CHARS[:36] for c in p):
            continue
        n = T.host_trie_walk(nodes, p)
        kids = trie.next_chars(p)
        if n < 0:
            assert not list(kids) and not trie.is_word(p)
            continue
        mask = int(nodes["mask"][n])
        want = 0
        for c in kids:
            if c != " " and c in synth.BASE_CHARS:
                want |= 1 << synth.BASE_CHARS.index(c)
        if trie.is_word(p) or " " in kids:
            want |= 1 << 36
        assert mask & ((1 << 37) - 1) == want, p
        assert bool((mask >> 37) & 1) == trie.is_word(p), p


def test_flat_trie_from_words_equals_from_objects():
    words = ["  Hello ", "help", "", None, "he", "h e"]
    a = T.flatten_trie(T.HostTrie.from_words(words))
    b = T.flatten_trie(O.build_trie(words))
    assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("order", [1, 2, 3, 5, 7, 10])
@pytest.mark.parametrize("floor", [True, False])
def test_ngram_tables_reproduce_logp_bit_exact(order, floor):
    words = synth.make_lexicon(60, 11, max_len=6)
    corpus = synth.make_sentences(words, 300, 5) + ["new-york 7", "ÉTÉ x", "a$b^c"]
    lm = O.NgramLM(order, floor_vsz=floor).fit(corpus)
    t = T.compile_ngram(lm)
    assert t.vsz == lm.vsz()
    for text in ["", " ", "a", "ab", "a ", "abc", "abcd", "abcde", "ab cd ", "zz9 q", "q", " ab"] + corpus[:25]:
        if any(c not in T.SYM for c in text.strip().lower()):
            continue
        assert T.host_logp(t, text).hex() == lm.logp(text).hex(), (order, floor, text)
        assert T.host_automaton_logp(t, text).hex() == lm.logp(text).hex(), (order, floor, text)


def test_ngram_tables_empty_lm_probe():
    class Bare:                                   # a reference-style object: no floor_vsz attribute
        def __init__(self, lm):
            self.order, self.counts, self.context, self.V, self.logp = lm.order, lm.counts, lm.context, lm.V, lm.logp

    for floor in (True, False):
        lm = O.NgramLM(3, floor_vsz=floor).fit([])
        t = T.compile_ngram(Bare(lm))
        assert T.host_logp(t, "abcd").hex() == lm.logp("abcd").hex()


def test_hash_tables_are_well_formed():
    lm = O.NgramLM(5).fit(synth.make_sentences(synth.make_lexicon(200, 0), 3000, 1))
    t = T.compile_ngram(lm)
    for tab, find in ((t.grams, T.host_find), (t.ctxs, T.host_find_ctx)):
        cap = tab.shape[0]
        assert cap & (cap - 1) == 0
        used = tab["key"] != T.KEY_EMPTY
        assert used.sum() * 2 <= cap
        keys = tab["key"][used]
        assert len(np.unique(keys)) == len(keys)
        for k in keys[:200]:
            assert find(tab, int(k)) >= 0
    # every context key sits in one of its two slots, and every seen context is in the table
    used = np.flatnonzero(t.ctxs["key"] != T.KEY_EMPTY)
    h = T.mix64(t.ctxs["key"][used])
    cap = np.uint64(t.ctxs.shape[0] - 1)
    assert (((h & cap) == used) | (((h >> np.uint64(32)) & cap) == used)).all()
    assert len(used) == len(t.ctx_recs) == len(lm.context)
    assert sorted(t.ctxs["ctx_id"][used].tolist()) == list(range(len(used)))


def test_two_choice_placement_under_pressure():
    rng = np.random.RandomState(5)
    keys = np.unique(rng.randint(0, 2 ** 62, size=40000).astype(np.uint64))
    cap = 1 << 17                                      # load 0.3: needs evictions, must succeed
    sl = T._place_two_choice(keys, cap)
    assert sl is not None and len(np.unique(sl)) == len(keys)
    h = T.mix64(keys)
    assert (((h & np.uint64(cap - 1)).astype(np.int64) == sl) | (((h >> np.uint64(32)) & np.uint64(cap - 1)).astype(np.int64) == sl)).all()
    assert T._place_two_choice(keys, 1 << 15) is None   # more keys than slots


def test_library_exports_every_declared_symbol():
    from silent_speech_amd import lib
    header = (ROOT / "include" / "nsdecode.h").read_text()
    declared = set(re.findall(r"\b(nsd_[a-z0-9_]+)\s*\(", header))
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    L = ctypes.CDLL(str(lib.SO_PATH))
    for name in declared:
        assert hasattr(L, name), name
    lib.load()
    assert lib.load().nsd_version() == 103
    assert b"beam" in lib.load().nsd_strerror(-2)
    # the exported slot hash equals the table builder's
    for k in (0, 1, 12345678901234567, 2 ** 64 - 2):
        assert lib.load().nsd_hash64(k) == int(T.mix64(np.array([k], dtype=np.uint64))[0])


def test_struct_layouts_match_header():
    from silent_speech_amd import lib
    assert ctypes.sizeof(lib.Params) == 6 * 4 + 4 * 8 + 8 * (128 + 4)
    assert ctypes.sizeof(lib.BeamStatePtrs) == 11 * 8
    assert ctypes.sizeof(lib.NgramHash) == 72
    assert T.CTXREC_DTYPE.itemsize == 32 and T.GRAMREC_DTYPE.itemsize == 24
    assert T.NODE_DTYPE.itemsize == 16 and T.GRAM_DTYPE.itemsize == 16 and T.CTX_DTYPE.itemsize == 32


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from silent_speech_amd import lib
    from silent_speech_amd.decoder import DeviceTables
    with pytest.raises(lib.NsdError):
        DeviceTables.from_words(["ab"])


def test_package_counts_and_text_generators_match_checker():
    from silent_speech_amd import synthetic as S
    words = S.make_lexicon(80, 3)
    assert words == synth.make_lexicon(80, 3, max_len=8)
    sents = S.make_sentences(words, 50, 4)
    assert sents == synth.make_sentences(words, 50, 4)
    for order in (1, 3, 5):
        a = T.NgramCounts(order).fit(sents + ["", " x ", "AB"])
        b = O.NgramLM(order).fit(sents + ["", " x ", "AB"])
        assert dict(a.counts) == dict(b.counts) and dict(a.context) == dict(b.context) and a.V == b.V
    tg, tl = S.pack_targets(["ab c", ""])
    assert tg.tolist() == [[0, 1, 36, 2], [0, 0, 0, 0]] and tl.tolist() == [4, 0]
