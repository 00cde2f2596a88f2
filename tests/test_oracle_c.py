"""The C checker (oracle/ns_oracle_c.c) against the Python restatement and torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cbind, synth
from oracle import ns_oracle as O
from tests.golden_util import case_corpus, case_targets, case_words, decode_kwargs, target_tokens


def test_det_log_softmax_close_to_torch_and_fp64():
    rng = np.random.RandomState(0)
    for scale in (0.5, 4.0, 30.0, 200.0):
        x = (rng.randn(4000, 40) * scale).astype(np.float32)
        x[::7, 3] += 60.0
        got = cbind.log_softmax_rows(x)
        ref32 = F.log_softmax(torch.from_numpy(x), dim=-1).numpy()
        ref64 = F.log_softmax(torch.from_numpy(x).double(), dim=-1).numpy()
        # absolute error of a few fp32 ulps of max(1, |value|): the same class of error
        # torch's own fp32 log_softmax has against the fp64 value
        tol = 3e-7 * np.maximum(1.0, np.abs(ref64))
        assert (np.abs(got.astype(np.float64) - ref64) <= tol).all()
        assert (np.abs(ref32.astype(np.float64) - ref64) <= tol).all()
        assert (np.abs(got.astype(np.float64) - ref32) <= 2 * tol).all()


def test_det_log_softmax_edge_rows():
    x = np.zeros((1, 40), dtype=np.float32)
    got = cbind.log_softmax_rows(x)[0]
    assert np.allclose(got, -np.log(40.0), atol=2e-7)
    x = np.full((1, 40), -1e4, dtype=np.float32); x[0, 5] = 1e4
    got = cbind.log_softmax_rows(x)[0]
    assert got[5] == 0.0 and got[0] == np.float32(-2e4)
    # other vocabulary sizes share the same tree
    for V in (1, 7, 8, 33, 64):
        x = np.random.RandomState(V).randn(1, V).astype(np.float32)
        got = np.array(cbind.log_softmax(x[0]))
        ref = F.log_softmax(torch.from_numpy(x).double(), dim=-1).numpy()[0]
        assert np.abs(got - ref).max() < 5e-7


def test_synth_row_matches_numpy():
    for u in range(5):
        useed = synth.utt_seed(77, u)
        assert useed == cbind.lib().nso_utt_seed(77, u)
        ids = [synth.BOS_ID, 3, 36, 7, 38][: 1 + u]
        tgt = [3, 5, 7]
        want = synth.logits_row(useed, ids, tgt, 12.5)
        step = len(ids) - 1
        tok = tgt[step] if step < len(tgt) else synth.EOS_ID
        got = cbind.synth_row(useed, synth.hist_of(ids), step, tok, 12.5)
        assert got.tobytes() == want.tobytes()


def test_lm_logp_matches_python(golden):
    words = synth.make_lexicon(60, 11, max_len=6)
    corpus = synth.make_sentences(words, 300, 5)
    for order in (1, 2, 3, 5, 7):
        for floor in (True, False):
            py = O.NgramLM(order, floor_vsz=floor).fit(corpus)
            cm = cbind.Model(words, corpus, order, floor)
            for text in ["", " ", "a", "ab", "a ", "abc", "abcd", "ab cd ", "zz9 q", "$", "^a"] + corpus[:20]:
                assert cm.logp(text).hex() == py.logp(text).hex(), (order, floor, text)
    py = O.NgramLM(3, floor_vsz=True).fit([])
    cm = cbind.Model([], [], 3, True)
    assert cm.logp("abcd").hex() == py.logp("abcd").hex()


def _ascii_case(case):
    return all(ord(ch) < 128 for w in case.get("extra_words", []) for ch in w)


def test_decode_matches_python_restatement(golden):
    """Same synthetic logits, deterministic log-softmax on both sides: ids,
    per-step parents/tokens/masks identical and scores bit-identical."""
    n = 0
    for case in golden["cases"]:
        words = case_words(case)
        corpus = case_corpus(case, words)
        trie = O.build_trie(words)
        floor = case["variant"] == "U"
        lm = O.NgramLM(case["lm_order"], floor_vsz=floor).fit(corpus) if case["lm_order"] > 0 else None
        cm = cbind.Model(words, corpus, case["lm_order"], floor)
        targets = case_targets(case, words)
        toks = [target_tokens(t) for t in targets]
        S = max(1, max(len(t) for t in toks))
        tg = np.zeros((len(toks), S), np.int8)
        for i, t in enumerate(toks):
            tg[i, :len(t)] = t
        kw = decode_kwargs(case)
        res = cm.decode_synth(variant=case["variant"], beam=case["beam"], max_new_tokens=case["max_new_tokens"],
                              beta=case["beta"], gamma=case["gamma"], kappa=case["kappa"], alpha=case["alpha"],
                              min_len_for_eos=case["min_len_for_eos"], margin=case["margin"],
                              base_seed=case["base_seed"], targets=tg,
                              target_lens=np.array([len(t) for t in toks], np.int32), trace=True)
        for u, tok in enumerate(toks):
            useed = synth.utt_seed(case["base_seed"], u)
            trace = []
            step_fn = lambda ids: synth.logits_row(useed, ids, tok, case["margin"])
            try:
                ids = O.constrained_beam_decode(step_fn, O.TextTransformLike, case["beam"], case["max_new_tokens"],
                                                trie, lm, trace=trace, log_softmax=O.lsm_det, **kw)
            except RuntimeError:
                assert res["lens"][u] == -1
                continue
            assert res["lens"][u] == len(ids)
            assert list(res["ids"][u, :len(ids)]) == ids, (case["name"], u)
            assert res["steps"][u] == len(trace)
            for t, st in enumerate(trace):
                nb = len(st["chosen"])
                assert res["nbeams"][u, t] == nb
                for r, (p, tk, sc) in enumerate(st["chosen"]):
                    assert res["parent"][u, t, r] == p and res["token"][u, t, r] == tk, (case["name"], u, t, r)
                    assert float(res["score"][u, t, r]).hex() == float(sc).hex(), (case["name"], u, t, r)
                for p, m in zip(st["live"], st["masks"]):
                    assert int(res["mask"][u, t, p]) == m
            n += 1
    assert n > 80


def test_ctc_greedy_matches_python():
    rng = np.random.RandomState(3)
    x = rng.randn(200, 40).astype(np.float32)
    x[np.arange(200), rng.randint(0, 40, 200) // 3 * 3 % 40] += 4.0
    x[10:20] = x[10]
    assert cbind.ctc_greedy(x) == O.ctc_greedy(x, O.TextTransformLike)
