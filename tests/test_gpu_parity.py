"""GPU parity: the HIP path (through the C ABI of libnsdecode.so) against the
CPU checkers and the golden vectors recorded from the reference.

Bars: trie masks, beam indices (parents/tokens/live histories), ids and
transcripts bit-exact; fp64 scores bit-exact against the checker that uses the
same deterministic log-softmax (and therefore within the 1e-5 relative
tolerance the task states against the reference's torch log-softmax)."""
import numpy as np
import pytest
import torch

from oracle import cbind, synth
from oracle import ns_oracle as O
from tests.golden_util import (case_corpus, case_model, case_targets, case_words, decode_kwargs, enc,
                               target_tokens)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _pack_targets(targets):
    toks = [target_tokens(t) for t in targets]
    S = max(1, max(len(t) for t in toks))
    tg = np.zeros((len(toks), S), np.int8)
    for i, t in enumerate(toks):
        tg[i, :len(t)] = t
    return tg, np.array([len(t) for t in toks], np.int32)


def test_log_softmax_bit_exact(dev):
    from silent_speech_amd.decoder import log_softmax_rows
    rng = np.random.RandomState(0)
    x = np.concatenate([(rng.randn(20000, 40) * s).astype(np.float32) for s in (0.3, 3.0, 30.0, 300.0)])
    x[::5, 7] += 60.0
    x[1] = 0.0
    x[2] = -1e30; x[2, 9] = 1e30
    got = log_softmax_rows(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = cbind.log_softmax_rows(x)
    assert got.tobytes() == want.tobytes()
    ref = torch.log_softmax(torch.from_numpy(x).double(), -1).numpy()
    assert (np.abs(got - ref) <= 3e-7 * np.maximum(1.0, np.abs(ref))).all()


def test_golden_cases_through_the_drop_in(golden, dev):
    """Every recorded reference run, replayed through the reference-signature
    entry point: same step_fn call sequence (= live beams in rank order at every
    step), same lexicon masks, same ids and transcript."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.decoder import constrained_beam_decode, ids_to_text
    code = golden["meta"]["token_code"]
    n = 0
    for case in golden["cases"]:
        words, trie, lm = case_model(case)
        for u, rec in enumerate(case["utts"]):
            useed = synth.utt_seed(case["base_seed"], u)
            tgt = target_tokens(rec["target"])
            calls = []

            def step_fn(ids):
                calls.append(enc(ids, code))
                return torch.from_numpy(synth.logits_row(useed, ids, tgt, case["margin"])).unsqueeze(0)

            kw = decode_kwargs(case)
            if "raises" in rec:
                with pytest.raises(RuntimeError):
                    constrained_beam_decode(step_fn, TextTransform, case["beam"], case["max_new_tokens"], trie, lm, **kw)
                assert calls == rec["calls"]
                continue
            ids = constrained_beam_decode(step_fn, TextTransform, case["beam"], case["max_new_tokens"], trie, lm, **kw)
            assert calls == rec["calls"], (case["name"], u)
            assert ids == rec["ids"], (case["name"], u)
            assert ids_to_text(TextTransform, ids) == rec["text"]
            n += 1
    assert n > 80


def test_golden_cases_through_the_per_file_bindings(golden, dev):
    """The same recorded runs through the bindings that carry each reference file's own signature, called the
    way the drivers call them (inference_emg_llm.py:432-435, eval_ar.py:533-545): keyword arguments, no variant
    switch."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.compat import eval_ar, inference_emg_llm, ns_utils
    mod = {"U": ns_utils, "I": inference_emg_llm, "E": eval_ar}
    code = golden["meta"]["token_code"]
    n = 0
    for case in golden["cases"]:
        words, trie, lm = case_model(case)
        for u, rec in enumerate(case["utts"]):
            useed = synth.utt_seed(case["base_seed"], u)
            tgt = target_tokens(rec["target"])
            calls = []

            def step_fn(ids):
                calls.append(enc(ids, code))
                return torch.from_numpy(synth.logits_row(useed, ids, tgt, case["margin"])).unsqueeze(0)

            kw = dict(beam=case["beam"], max_new_tokens=case["max_new_tokens"], trie=trie, lm=lm,
                      lm_weight=case["beta"], oov_penalty=case["gamma"])
            if case["variant"] == "E":
                kw.update(word_bonus=case["kappa"], length_penalty_alpha=case["alpha"],
                          min_len_for_eos=case["min_len_for_eos"])
            m = mod[case["variant"]]
            if "raises" in rec:
                with pytest.raises(RuntimeError):
                    m.constrained_beam_decode(step_fn, TextTransform, **kw)
                continue
            ids = m.constrained_beam_decode(step_fn, TextTransform, **kw)
            assert calls == rec["calls"], (case["name"], u)
            assert ids == rec["ids"], (case["name"], u)
            assert m.ids_to_text(TextTransform, ids) == rec["text"]
            n += 1
    assert n > 80


def test_drop_in_notices_a_refitted_language_model(dev):
    """The drop-in caches device tables per (trie, lm) object; refitting the same LM object must not decode with
    the old tables."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.compat import ns_utils
    words = ["ab", "ba"]
    trie = O.build_trie(words)
    lm = O.NgramLM(2).fit(["ab ab ab ab"])
    flat = lambda ids: torch.zeros(1, 40)                              # noqa: E731
    a = ns_utils.constrained_beam_decode(flat, TextTransform, 2, 3, trie, lm, lm_weight=5.0, allow_eos_anytime=False)
    want_a = O.constrained_beam_decode(flat, O.TextTransformLike, 2, 3, trie, lm, lm_weight=5.0, allow_eos_anytime=False)
    lm.fit(["ba ba ba ba ba ba ba ba ba ba"])
    b = ns_utils.constrained_beam_decode(flat, TextTransform, 2, 3, trie, lm, lm_weight=5.0, allow_eos_anytime=False)
    want_b = O.constrained_beam_decode(flat, O.TextTransformLike, 2, 3, trie, lm, lm_weight=5.0, allow_eos_anytime=False)
    assert a == want_a and b == want_b and want_a != want_b


def test_drop_in_notices_a_same_size_refit_and_a_trie_insert(dev):
    """A refit that changes counts but neither the set of n-grams nor any dictionary size, and a word inserted
    under an existing first letter, both lead to new device tables (the reference reads the live objects)."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.compat import ns_utils
    words = ["ab", "ba"]
    trie = O.build_trie(words)
    lm = O.NgramLM(2).fit(["ab ba ab ab ab ab ab"])
    flat = lambda ids: torch.zeros(1, 40)                              # noqa: E731
    run = lambda: ns_utils.constrained_beam_decode(flat, TextTransform, 2, 4, trie, lm, lm_weight=5.0,   # noqa: E731
                                                   allow_eos_anytime=False)
    want = lambda: O.constrained_beam_decode(flat, O.TextTransformLike, 2, 4, trie, lm, lm_weight=5.0,   # noqa: E731
                                             allow_eos_anytime=False)
    a, want_a = run(), want()
    sizes = (len(lm.counts), len(lm.context), len(lm.V))
    lm.fit(["ab ba ba ba ba ba ab"])
    assert (len(lm.counts), len(lm.context), len(lm.V)) == sizes
    b, want_b = run(), want()
    assert a == want_a and b == want_b and want_a != want_b
    lm.fit(["aa"] * 30)                                                # (fit adds to the counts it has)
    c, want_c = run(), want()
    trie.insert("aa")                                                  # the root's children stay {a, b}
    d, want_d = run(), want()
    assert c == want_c and d == want_d and want_d != want_c


def test_drop_in_rejects_a_model_without_counts(dev):
    """A KenLM-style object (only .logp) is refused by name, before anything is decoded."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.compat import ns_utils

    class OnlyLogp:
        def logp(self, s):
            return 0.0
    with pytest.raises(TypeError, match="KenLM"):
        ns_utils.constrained_beam_decode(lambda ids: torch.zeros(1, 40), TextTransform, 2, 3, O.build_trie(["ab"]),
                                         OnlyLogp())


def _run_case_batched(case, dev, with_masks=True):
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    words = case_words(case)
    corpus = case_corpus(case, words)
    floor = case["variant"] == "U"
    lm = O.NgramLM(case["lm_order"], floor_vsz=floor).fit(corpus) if case["lm_order"] > 0 else None
    targets = case_targets(case, words)
    tg, tl = _pack_targets(targets)
    tables = DeviceTables.from_words(words, lm)
    dec = NSDecoder(tables, case["beam"], case["max_new_tokens"], variant=case["variant"], lm_weight=case["beta"],
                    oov_penalty=case["gamma"], word_bonus=case["kappa"], length_penalty_alpha=case["alpha"],
                    min_len_for_eos=case["min_len_for_eos"], lanes_per_utt=case.get("lanes", 0),
                    allow_eos_anytime=case.get("allow_eos_anytime", True))
    ws = dec.workspace(len(targets), with_masks=with_masks)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    err = None
    try:
        ids, lens, scores = dec.decode_synth(tgd, tld, base_seed=case["base_seed"], margin=case["margin"], ws=ws,
                                             check_every=1)
    except RuntimeError as e:
        err = e
        ids = lens = scores = None
    cm = cbind.Model(words, corpus, case["lm_order"], floor)
    ref = cm.decode_synth(variant=case["variant"], beam=case["beam"], max_new_tokens=case["max_new_tokens"],
                          beta=case["beta"], gamma=case["gamma"], kappa=case["kappa"], alpha=case["alpha"],
                          min_len_for_eos=case["min_len_for_eos"], margin=case["margin"],
                          allow_eos_anytime=case.get("allow_eos_anytime", True),
                          base_seed=case["base_seed"], targets=tg, target_lens=tl, trace=True)
    return ws, ids, lens, scores, ref, err


def _compare_with_trace(ws, ids, lens, scores, ref, name):
    N, B, T = ws.N, ws.B, ws.T
    ids, lens, scores = ids.cpu().numpy(), lens.cpu().numpy(), scores.cpu().numpy()
    bp = ws.backptr.cpu().numpy().reshape(max(T, 1), N, B)
    tk = ws.token.cpu().numpy().reshape(max(T, 1), N, B)
    utt = ws.utt.cpu().numpy()
    masks = ws.masks.cpu().numpy().reshape(max(T, 1), N, B).view(np.uint64) if ws.masks is not None else None
    assert (lens == ref["lens"]).all(), name
    for u in range(N):
        assert (ids[u, :lens[u]] == ref["ids"][u, :lens[u]]).all(), (name, u)
        assert float(scores[u]).hex() == float(ref["scores"][u]).hex(), (name, u)
        steps = int(utt[u]) >> 16
        assert steps == ref["steps"][u], (name, u)
        for t in range(steps):
            nb = ref["nbeams"][u, t]
            want_tok = ref["token"][u, t, :nb].astype(np.int16)
            got_tok = tk[t, u, :nb].astype(np.int16)
            got_tok[got_tok == 0xFF] = -1
            assert (bp[t, u, :nb] == ref["parent"][u, t, :nb]).all(), (name, u, t)
            assert (got_tok == want_tok).all(), (name, u, t)
            if masks is not None:
                nprev = 1 if t == 0 else ref["nbeams"][u, t - 1]
                assert (masks[t, u, :nprev] == ref["mask"][u, t, :nprev]).all(), (name, u, t)


def test_golden_cases_batched_against_c_checker(golden, dev):
    """All utterances of a case in one batch with the device-side synthetic AR
    head; parents, tokens, masks per step and fp64 scores against the C checker."""
    for case in golden["cases"]:
        if any(ord(ch) > 127 for w in case.get("extra_words", []) for ch in w):
            continue
        ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
        if err is not None:
            assert (ref["lens"] == -1).any(), case["name"]
            continue
        _compare_with_trace(ws, ids, lens, scores, ref, case["name"])
        # final ids equal the reference's own recorded output
        for u, rec in enumerate(case["utts"]):
            if "ids" in rec:
                assert list(ids[u, :int(lens[u])].cpu().numpy()) == rec["ids"], (case["name"], u)


@pytest.mark.parametrize("variant,beam,order,beta,margin", [
    ("U", 16, 5, 0.05, 60.0), ("I", 16, 5, 0.05, 60.0), ("E", 16, 5, 0.05, 60.0),
    ("U", 32, 5, 0.05, 8.0), ("I", 8, 3, 0.3, 6.0), ("E", 32, 5, 0.1, 10.0), ("U", 2, 2, 0.2, 5.0),
    ("I", 64, 5, 0.05, 5.0), ("U", 8, 1, 0.3, 6.0), ("E", 16, 1, 0.05, 20.0), ("I", 5, 7, 1.0, 2.0),
])
def test_batch_512_against_c_checker(dev, variant, beam, order, beta, margin):
    case = dict(variant=variant, beam=beam, lm_order=order, beta=beta, margin=margin, gamma=1.0, kappa=0.35,
                alpha=0.6, min_len_for_eos=8, max_new_tokens=64 if beam <= 16 else 40, n_words=200, lex_seed=0,
                max_len=8, n_corpus=4000, corpus_seed=1, n_utts=512 if beam <= 32 else 64, target_seed=2,
                base_seed=99, min_words=3, max_words=7, name=f"b512_{variant}_{beam}")
    if variant != "U" and beam > 40:
        case["beam"] = 40
    ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
    assert err is None
    _compare_with_trace(ws, ids, lens, scores, ref, case["name"])


def test_open_lexicon_with_dead_beams(dev):
    """100k-word lexicon over a-z0-9 plus apostrophe words whose prefix node has
    no usable child: exercises random-access trie reads and the fall-back."""
    # apostrophe words with a prefix no random word shares: at that prefix node the only child is "'",
    # which the decoder cannot take, so the hypothesis falls back to the neural top-k and stays dead
    words = synth.make_lexicon(100000, 1, synth.BASE_CHARS[:36], 2, 12) + ["zqxjkvw'a", "wvvqkzx'b", "it's"]
    case = dict(variant="U", beam=32, lm_order=5, beta=0.05, margin=9.0, gamma=1.0, kappa=0.35, alpha=0.6,
                min_len_for_eos=8, max_new_tokens=24, n_words=0, lex_seed=0, n_corpus=0, corpus_seed=0,
                n_utts=0, target_seed=0, base_seed=5, name="open100k")
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    plain = [w for w in words if "'" not in w]
    corpus = synth.make_sentences(plain, 3000, 4)
    targets = synth.make_sentences(plain, 192, 6, 2, 4) + ["zqxjkvw ab it s", "wvvqkzx b c", "it sx"] * 8
    tg, tl = _pack_targets(targets)
    saw_dead = False
    for variant in ("U", "I", "E"):
        floor = variant == "U"
        lm = O.NgramLM(5, floor_vsz=floor).fit(corpus)
        tables = DeviceTables.from_words(words, lm)
        dec = NSDecoder(tables, 32, 24, variant=variant, lm_weight=0.05, oov_penalty=1.0, word_bonus=0.35)
        ws = dec.workspace(len(targets), with_masks=True)
        ids, lens, scores = dec.decode_synth(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev),
                                             base_seed=5, margin=9.0, ws=ws, check_every=4)
        ref = cbind.Model(words, corpus, 5, floor).decode_synth(
            variant=variant, beam=32, max_new_tokens=24, beta=0.05, gamma=1.0, kappa=0.35, margin=9.0,
            base_seed=5, targets=tg, target_lens=tl, trace=True)
        _compare_with_trace(ws, ids, lens, scores, ref, f"open100k_{variant}")
        # the fall-back ran if some parent that was not finished had an empty lexicon mask
        tk = ws.token.cpu().numpy().reshape(24, len(targets), 32)
        saw_dead = saw_dead or bool(((ws.meta.cpu().numpy() & 2) != 0).any()) or bool((tk[:, :, :] == 37).any())
    assert saw_dead, "expected the fall-back branch to run in this workload"


def test_edge_shapes(dev):
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    tables = DeviceTables.from_words(["ab", "abc", "b"], O.NgramLM(3).fit(["ab abc b"]))
    # zero utterances, zero steps
    dec = NSDecoder(tables, 4, 8)
    ids, lens, _ = dec.decode_synth(torch.zeros(0, 1, dtype=torch.int8, device=dev),
                                    torch.zeros(0, dtype=torch.int32, device=dev), base_seed=1, margin=5.0)
    assert ids.shape[0] == 0 and lens.shape[0] == 0
    dec0 = NSDecoder(tables, 4, 0)
    ids, lens, _ = dec0.decode_synth(torch.zeros(3, 1, dtype=torch.int8, device=dev),
                                     torch.zeros(3, dtype=torch.int32, device=dev), base_seed=1, margin=5.0)
    assert lens.tolist() == [1, 1, 1] and ids[:, 0].tolist() == [37, 37, 37]
    # maximum steps and beam, ragged targets (empty target included)
    dec = NSDecoder(tables, 64, 128, variant="U", lm_weight=0.01)
    tg = np.zeros((5, 40), np.int8); tl = np.array([0, 1, 7, 40, 3], np.int32)
    tg[:] = np.tile(np.array([0, 1, 36, 1, 36, 0, 1, 2, 36, 1], np.int8), 4)
    ws = dec.workspace(5, with_masks=True)
    ids, lens, scores = dec.decode_synth(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev),
                                         base_seed=3, margin=40.0, ws=ws)
    ref = cbind.Model(["ab", "abc", "b"], ["ab abc b"], 3, True).decode_synth(
        variant="U", beam=64, max_new_tokens=128, beta=0.01, gamma=1.0, margin=40.0, base_seed=3,
        targets=tg, target_lens=tl, trace=True)
    _compare_with_trace(ws, ids, lens, scores, ref, "edge")
    with pytest.raises(ValueError):
        NSDecoder(tables, 65, 8)
    with pytest.raises(ValueError):
        NSDecoder(tables, 4, 129)
    # nsd_step indexes hypotheses and logits rows in 32 bits: a launch with N * B beyond NSD_MAX_HYPS is refused
    # before anything is enqueued (the pointers are never used)
    import ctypes as C
    from silent_speech_amd import lib as _lib
    L = _lib.load()
    small = dec.workspace(2)
    too_many = (1 << 26) // 64 + 1
    rc = L.nsd_step(tables.handle, C.byref(dec.params), small.logits.data_ptr(), C.byref(small.ptrs), C.byref(small.ptrs),
                    small.backptr.data_ptr(), small.token.data_ptr(), None, too_many, 64, 40, None)
    assert rc == -7 and b"size" in L.nsd_strerror(rc).lower()


def test_ctc_greedy(dev):
    from silent_speech_amd.decoder import ctc_greedy_batch
    rng = np.random.RandomState(2)
    N, Tp = 37, 256
    x = rng.randn(N, Tp, 40).astype(np.float32)
    # planted frame-repeated targets with blanks (PAD is the CTC blank)
    for u in range(N):
        t = 0
        while t < Tp:
            tok = int(rng.choice([rng.randint(0, 37), 39, 39, 37, 38]))
            run = int(rng.randint(1, 6))
            x[u, t:t + run, tok] += 6.0
            t += run
    x[3] = 0.0                                     # all-ties rows: first index wins
    ids, lens = ctc_greedy_batch(torch.from_numpy(x).to(dev))
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    for u in range(N):
        want = O.ctc_greedy(x[u], O.TextTransformLike)
        assert list(ids[u, :lens[u]]) == want, u
    # ragged T' that is not a multiple of 64, and the known answers of the collapse rule
    for Tp in (1, 63, 65, 130):
        y = rng.randn(5, Tp, 40).astype(np.float32)
        ids, lens = ctc_greedy_batch(torch.from_numpy(y).to(dev))
        for u in range(5):
            assert list(ids[u, :int(lens[u])].cpu().numpy()) == cbind.ctc_greedy(y[u])
    z = np.full((1, 3, 40), -5.0, np.float32); z[0, 0, 0] = z[0, 1, 39] = z[0, 2, 0] = 3.0
    ids, lens = ctc_greedy_batch(torch.from_numpy(z).to(dev))
    assert list(ids[0, :int(lens[0])].cpu().numpy()) == [0, 0]


def test_lm_logp_and_rerank(dev):
    from silent_speech_amd.decoder import DeviceTables, lm_logp_batch, rerank_gather_batch
    words = synth.make_lexicon(60, 11, max_len=6)
    corpus = synth.make_sentences(words, 300, 5)
    texts = ["", " ", "a", "ab", "a ", "abc", "abcd", "ab cd ", " ab", "zz9 q"] + corpus[:40]
    S = max(len(t) for t in texts)
    for order in (1, 2, 3, 5, 7):
        lm = O.NgramLM(order).fit(corpus)
        tables = DeviceTables.from_words(words, lm)
        ids = np.zeros((len(texts), S), np.uint8)
        lens = np.array([len(t) for t in texts], np.int32)
        for i, t in enumerate(texts):
            ids[i, :len(t)] = synth.text_to_tokens(t)
        got = lm_logp_batch(tables, torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
        for i, t in enumerate(texts):
            assert float(got[i]).hex() == lm.logp(t).hex(), (order, t)
        # LM rerank rule (inference_emg_llm.py:461-464) on device scores
        if order == 3:
            a, b = 12, 20
            f = lambda i: float(got[i]) - 0.002 * len(texts[i])
            pick = texts[a] if f(a) >= f(b) else texts[b]
            assert pick == O.rerank_lm(texts[a], texts[b], lm)
    # teacher-forced token log-prob sums (eval_ar.py:574-589)
    rng = np.random.RandomState(4)
    N, Lm = 33, 50
    lg = (rng.randn(N, Lm, 40) * 3).astype(np.float32)
    tgt = rng.randint(0, 40, (N, Lm)).astype(np.uint8)
    ln = rng.randint(0, Lm + 1, N).astype(np.int32)
    got = rerank_gather_batch(torch.from_numpy(lg).to(dev), torch.from_numpy(tgt).to(dev),
                              torch.from_numpy(ln).to(dev)).cpu().numpy()
    lsm = cbind.log_softmax_rows(lg)
    for u in range(N):
        tot = 0.0
        for i in range(ln[u]):
            tot += float(lsm[u, i, tgt[u, i]])
        assert float(got[u]).hex() == tot.hex(), u


def test_lm_scores_of_long_strings_on_both_kernel_forms(dev):
    """Strings longer than one round of 64 windows, with leading / trailing / inner spaces, through nsd_lm_logp and
    nsd_rerank_lm: the wave-per-string kernels (strings staged in LDS) and the one-lane-per-string kernels the
    library takes beyond 16 384 bytes per row; bit-equal to CharNgramLM.logp."""
    from silent_speech_amd import TextTransform as TT
    from silent_speech_amd.decoder import DeviceTables, lm_logp_batch, rerank_lm_batch
    words = synth.make_lexicon(80, 12, max_len=7)
    corpus = synth.make_sentences(words, 400, 6)
    rng = np.random.RandomState(8)
    texts = []
    for n in (63, 64, 65, 127, 128, 129, 200, 333):
        t = " ".join(corpus[int(i)] for i in rng.randint(0, len(corpus), 40))
        texts += [t[:n], "  " + t[:n - 4] + "  ", t[3:3 + n]]
    texts += ["", "   ", "a"]
    for order in (1, 3, 5):
        lm = O.NgramLM(order).fit(corpus)
        tables = DeviceTables.from_words(words, lm)
        for S in (max(len(t) for t in texts), 16400):
            ids = np.zeros((len(texts), S), np.uint8)
            lens = np.array([len(t) for t in texts], np.int32)
            for i, t in enumerate(texts):
                ids[i, :len(t)] = synth.text_to_tokens(t)
            got = lm_logp_batch(tables, torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)).cpu().numpy()
            for i, t in enumerate(texts):
                assert float(got[i]).hex() == lm.logp(t).hex(), (order, S, i)
            # the same strings as decoded ids: BOS first, an EOS and a PAD inside, garbage after the EOS
            ar = np.full((len(texts), S + 8), TT.PAD_IDX, np.uint8)
            al = np.zeros(len(texts), np.int32)
            for i, t in enumerate(texts):
                row = [TT.BOS_IDX] + synth.text_to_tokens(t)[:len(t) // 2] + [TT.PAD_IDX] + \
                      synth.text_to_tokens(t)[len(t) // 2:] + [TT.EOS_IDX, 3, 4]
                ar[i, :len(row)] = row
                al[i] = len(row)
            ctc = ids.copy()
            choice, scores = rerank_lm_batch(tables, torch.from_numpy(ar).to(dev), torch.from_numpy(al).to(dev),
                                             torch.from_numpy(ctc).to(dev), torch.from_numpy(lens).to(dev))
            choice, scores = choice.cpu().numpy(), scores.cpu().numpy()
            for i, t in enumerate(texts):
                want = float(lm.logp(t.strip())) - 0.002 * len(t.strip())
                assert float(scores[i, 0]).hex() == want.hex() and float(scores[i, 1]).hex() == want.hex(), (order, S, i)
                assert choice[i] == 0                                   # equal scores: the AR transcript stays


def test_tables_with_a_value_that_is_not_finite_are_refused(dev):
    """The bounds that drop candidates before their look-up rest on the range of the table values: a NaN or an
    infinity in any log-prob is an error of nsd_tables_create, not a silent mis-prune."""
    from silent_speech_amd import lib
    from silent_speech_amd.decoder import DeviceTables
    from silent_speech_amd.tables import HostTrie, compile_ngram, flatten_trie
    words = ["ab", "ba"]
    lm = O.NgramLM(3).fit(["ab ba ab", "ba ba"])
    nodes = flatten_trie(HostTrie.from_words(words))
    for field, bad in (("gram_recs", np.nan), ("ctxs", np.inf), ("grams", -np.inf)):
        ng = compile_ngram(lm, None)
        arr = getattr(ng, field)
        name = "logp" if "logp" in arr.dtype.names else "term"
        arr[name][np.flatnonzero(arr["key"] != np.uint64(0xFFFFFFFFFFFFFFFF))[0] if "key" in arr.dtype.names else 0] = bad
        with pytest.raises(lib.NsdError, match="malformed table"):
            DeviceTables(nodes, ng)
    DeviceTables(nodes, compile_ngram(lm, None))                        # the untouched tables are fine


def test_ar_step_kernels_refuse_misaligned_rows(dev):
    from silent_speech_amd import lib
    L = lib.load()
    x = torch.zeros(4 * 64 + 8, dtype=torch.bfloat16, device=dev)
    w = torch.ones(64, dtype=torch.bfloat16, device=dev)
    h = torch.zeros(4 * 64, dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert L.nsd_add_rmsnorm(x.data_ptr(), None, w.data_ptr(), 1e-5, 4, 64, 1, None, h.data_ptr(), st) == 0
    assert L.nsd_add_rmsnorm(x.data_ptr() + 2, None, w.data_ptr(), 1e-5, 4, 64, 1, None, h.data_ptr(), st) == lib.E_ALIGN
    g = torch.zeros(4 * 128 + 8, dtype=torch.bfloat16, device=dev)
    assert L.nsd_swiglu(g.data_ptr(), 4, 64, 1, h.data_ptr(), st) == 0
    assert L.nsd_swiglu(g.data_ptr() + 2, 4, 64, 1, h.data_ptr(), st) == lib.E_ALIGN
    assert b"16 bytes" in L.nsd_strerror(lib.E_ALIGN)
    torch.cuda.synchronize()


def test_gather_ids_matches_host_histories(dev):
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    words = synth.make_lexicon(50, 1)
    lm = O.NgramLM(5).fit(synth.make_sentences(words, 200, 2))
    targets = synth.make_sentences(words, 9, 3, 2, 3)
    tg, tl = _pack_targets(targets)
    dec = NSDecoder(DeviceTables.from_words(words, lm), 8, 20, variant="I", lm_weight=0.05)
    ws = dec.workspace(len(targets))
    dec.init(ws)
    hyps = [[[37]] for _ in targets]
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    for t in range(12):
        dec.synth_logits(ws, 11, 0, tgd, tld, 7.0)
        dec.step(ws)
        bp = ws.backptr[t].cpu().numpy().reshape(len(targets), 8)
        tk = ws.token[t].cpu().numpy().reshape(len(targets), 8)
        nb = ws.utt.cpu().numpy() & 0xFF
        for u in range(len(targets)):
            hyps[u] = [hyps[u][bp[u, r]] + ([] if tk[u, r] == 0xFF else [int(tk[u, r])]) for r in range(nb[u])]
    ids, lens = dec.gather_ids(ws)
    ids, lens = ids.cpu().numpy().reshape(len(targets), 8, -1), lens.cpu().numpy().reshape(len(targets), 8)
    for u in range(len(targets)):
        for r in range(8):
            if r < len(hyps[u]):
                assert list(ids[u, r, :lens[u, r]]) == hyps[u][r]
            else:
                assert lens[u, r] == 0


def test_native_decode_loop_equals_python_driven_loop(dev):
    """nsd_decode_synth (whole loop enqueued by native code, with and without the done poll and the
    per-launch events) gives the same ids, scores and back-pointers as the step-by-step calls."""
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    words = synth.make_lexicon(200, 0, max_len=8)
    corpus = synth.make_sentences(words, 4000, 1)
    targets = synth.make_sentences(words, 300, 7, 3, 7)
    tg, tl = _pack_targets(targets)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    for variant in ("U", "I", "E"):
        lm = O.NgramLM(5, floor_vsz=(variant == "U")).fit(corpus)
        dec = NSDecoder(DeviceTables.from_words(words, lm), 16, 48, variant=variant, lm_weight=0.05, word_bonus=0.35)
        ws0 = dec.workspace(len(targets))
        ids0, lens0, sc0 = dec.decode_synth(tgd, tld, base_seed=3, margin=30.0, ws=ws0, check_every=0)
        for check_every, timed in ((0, False), (8, False), (0, True), (5, True)):
            ws1 = dec.workspace(len(targets))
            sm = np.zeros(48, np.float32) if timed else None
            ids1, lens1, sc1 = dec.decode_synth_native(tgd, tld, base_seed=3, margin=30.0, ws=ws1,
                                                       check_every=check_every, step_ms=sm)
            torch.cuda.synchronize()
            assert torch.equal(lens0, lens1) and torch.equal(ids0, ids1) and torch.equal(sc0, sc1)
            n = ws1.steps_run
            assert torch.equal(ws0.backptr[:n], ws1.backptr[:n]) and torch.equal(ws0.token[:n], ws1.token[:n])
            if timed:
                assert (sm[:n] > 0).all() and (sm[n:] == -1).all()


def test_exact_score_ties_follow_the_canonical_order(dev):
    """Uniform logits and no LM make every candidate of a hypothesis tie exactly; pruning must then keep
    the first candidates in emission order (parent rank, then token id), as a stable sort does."""
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    words = synth.make_lexicon(300, 2, max_len=5)
    trie = O.build_trie(words)
    for variant, B in (("I", 4), ("I", 16), ("U", 8), ("E", 16), ("I", 40)):
        dec = NSDecoder(DeviceTables.from_words(words, None), B, 10, variant=variant, oov_penalty=1.0,
                        word_bonus=0.0, min_len_for_eos=3)
        ws = dec.workspace(3, with_masks=True)
        dec.init(ws)
        hyps = [[(0.0, [37], "")] for _ in range(3)]
        for t in range(10):
            dec.step(ws, torch.zeros(3 * B, 40, dtype=torch.float32, device=dev))
            bp = ws.backptr[t].cpu().numpy().reshape(3, B)
            tk = ws.token[t].cpu().numpy().reshape(3, B)
            nb = ws.utt.cpu().numpy() & 0xFF
            # the same step in the Python restatement, fed the same all-zero logits
            trace = []
            want = O.constrained_beam_decode(lambda ids: np.zeros(40, np.float32), O.TextTransformLike, B, t + 1,
                                             trie, None, oov_penalty=1.0, variant=variant, word_bonus=0.0,
                                             min_len_for_eos=3, log_softmax=O.lsm_det, trace=trace)
            chosen = trace[-1]["chosen"] if len(trace) == t + 1 else None
            if chosen is None:
                break
            for u in range(3):
                assert nb[u] == len(chosen)
                for r, (p, tok, sc) in enumerate(chosen):
                    assert bp[u, r] == p and (tk[u, r] == (0xFF if tok < 0 else tok)), (variant, B, t, u, r)


def test_ctc_proposal_and_lm_rerank_pipeline(dev):
    """The CTC + rerank configuration: AR decode (variant I, as inference_emg_llm.py calls it) + CTC greedy proposal +
    LM rerank, all on device, against the Python restatement of inference_emg_llm.py:436-464 without
    postprocess."""
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, ctc_greedy_batch, rerank_lm_batch
    TT = O.TextTransformLike
    words = synth.make_lexicon(200, 0, max_len=8)
    corpus = synth.make_sentences(words, 4000, 1)
    targets = synth.make_sentences(words, 257, 8, 2, 5)
    tg, tl = _pack_targets(targets)
    lm = O.NgramLM(3, floor_vsz=False).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    dec = NSDecoder(tables, 16, 24, variant="I", lm_weight=0.3, oov_penalty=1.0)
    ar_ids, ar_len, _ = dec.decode_synth_native(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev),
                                                base_seed=4, margin=20.0)
    # CTC logits: the target with frame repeats and blanks, a few corrupted frames, some empty proposals
    rng = np.random.RandomState(9)
    Tp = 96
    x = rng.randn(len(targets), Tp, 40).astype(np.float32)
    for u, t in enumerate(targets):
        toks = synth.text_to_tokens(t)[: Tp // 4]
        f = 0
        for tok in toks:
            x[u, f:f + 2, tok] += 8.0; x[u, f + 2, 39] += 8.0; f += 3
        x[u, f:, 39] += 8.0
        if u % 5 == 0:
            x[u, rng.randint(0, max(f, 1)), rng.randint(0, 36)] += 20.0
        if u % 17 == 0:
            x[u] = 0.0; x[u, :, 39] = 5.0                      # empty CTC proposal
        if u % 19 == 0:
            x[u, :3, 36] += 30.0                               # leading spaces -> stripped
    ctc_ids, ctc_len = ctc_greedy_batch(torch.from_numpy(x).to(dev))
    choice, scores = rerank_lm_batch(tables, ar_ids, ar_len, ctc_ids, ctc_len)
    ar_np, al = ar_ids.cpu().numpy(), ar_len.cpu().numpy()
    choice, scores = choice.cpu().numpy(), scores.cpu().numpy()
    n_ctc = 0
    for u in range(len(targets)):
        hyp_ar = O.ids_to_text(TT, ar_np[u, :al[u]].tolist())
        hyp_ctc = O.ctc_hypothesis(x[u], TT)
        want = O.rerank_lm(hyp_ar, hyp_ctc, lm)
        got = hyp_ctc if choice[u] else hyp_ar
        assert got == want, (u, hyp_ar, hyp_ctc)
        assert float(scores[u, 0]).hex() == (float(lm.logp(hyp_ar)) - 0.002 * len(hyp_ar)).hex()
        assert float(scores[u, 1]).hex() == (float(lm.logp(hyp_ctc)) - 0.002 * len(hyp_ctc)).hex()
        n_ctc += int(choice[u])
    assert 0 < n_ctc < len(targets)
    # the driver's whole tail, postprocess included (inference_emg_llm.py:436-464)
    from silent_speech_amd.decoder import transcribe_batch
    from silent_speech_amd import TextTransform
    trie = O.build_trie(words)
    final = transcribe_batch(TextTransform, ar_ids, ar_len, tables, trie, words, ctc_logits=torch.from_numpy(x).to(dev))
    plain = transcribe_batch(TextTransform, ar_ids, ar_len, tables, trie.is_word, words)
    for u in range(len(targets)):
        hyp_ar = O.postprocess(O.ids_to_text(TT, ar_np[u, :al[u]].tolist()), trie, words)
        hyp_ctc = O.postprocess(O.ctc_hypothesis(x[u], TT), trie, words)
        assert plain[u] == hyp_ar, u
        assert final[u] == O.rerank_lm(hyp_ar, hyp_ctc, lm), u


@pytest.mark.gpu
def test_joint_rerank_tail_of_the_evaluation_driver(dev):
    """eval_ar.py:546-596 for a batch: AR decode (variant E), CTC proposal, postprocess of both, then the choice by
    teacher-forced AR log-prob + beta * LM log-prob.  The AR head is the synthetic one (a function of the token
    history), evaluated on the host for every prefix of both candidates; checked against the restatement with the
    same deterministic log-softmax (choice identical, scores to 1e-12 relative)."""
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, transcribe_batch_joint
    from silent_speech_amd import TextTransform
    TT = O.TextTransformLike
    words = synth.make_lexicon(120, 3, max_len=7)
    corpus = synth.make_sentences(words, 3000, 1)
    targets = synth.make_sentences(words, 48, 21, 2, 4)
    tg, tl = _pack_targets(targets)
    lm = O.NgramLM(3, floor_vsz=False).fit(corpus)
    tables = DeviceTables.from_words(words, lm)
    beta, margin, seed = 0.3, 6.0, 12
    dec = NSDecoder(tables, 8, 32, variant="E", lm_weight=beta, oov_penalty=1.0, word_bonus=0.35)
    ar_ids, ar_len, _ = dec.decode_synth_native(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev),
                                                base_seed=seed, margin=margin)
    rng = np.random.RandomState(2)
    Tp = 120
    x = rng.randn(len(targets), Tp, 40).astype(np.float32)
    for u, t in enumerate(targets):
        toks = synth.text_to_tokens(t)[: Tp // 3]
        if u % 4 == 0 and len(toks) > 3:
            toks[2] = (toks[2] + 1) % 26                              # a wrong letter in the CTC stream
        f = 0
        for tok in toks:
            x[u, f:f + 2, tok] += 8.0; x[u, f + 2, 39] += 8.0; f += 3
        x[u, f:, 39] += 8.0
    N = len(targets)
    tgt_tokens = [synth.text_to_tokens(t) for t in targets]

    def step_row(u, ids):
        return synth.logits_row(synth.utt_seed(seed, u), ids, tgt_tokens[u], margin)

    def teacher_logits(seq, lens):
        sq, ln = seq.cpu().numpy(), lens.cpu().numpy()
        out = np.zeros((sq.shape[0], sq.shape[1] - 1, 40), np.float32)
        for m in range(sq.shape[0]):
            for i in range(1, ln[m]):
                out[m, i - 1] = step_row(m % N, sq[m, :i].tolist())
        return torch.from_numpy(out).to(seq.device)

    trie = O.build_trie(words)
    final, scores = transcribe_batch_joint(TextTransform, ar_ids, ar_len, tables, trie, words,
                                           torch.from_numpy(x).to(dev), teacher_logits, beta)
    ar_np, al = ar_ids.cpu().numpy(), ar_len.cpu().numpy()
    n_ctc = 0
    for u in range(N):
        hyp_ar = O.postprocess(O.ids_to_text(TT, ar_np[u, :al[u]].tolist()), trie, words)
        hyp_ctc = O.postprocess(O.ctc_hypothesis(x[u], TT), trie, words)
        step_fn = lambda ids, u=u: step_row(u, ids)                  # noqa: E731
        want = O.rerank_joint(hyp_ar, hyp_ctc, step_fn, TT, lm, beta, O.lsm_det)
        assert final[u] == want, (u, hyp_ar, hyp_ctc, scores[u])
        for c, sxt in ((0, hyp_ar), (1, hyp_ctc)):
            ref = O.ar_logprob(sxt, step_fn, TT, O.lsm_det) + beta * float(lm.logp(sxt))
            assert abs(scores[u, c] - ref) <= 1e-12 * max(1.0, abs(ref)), (u, c)
        n_ctc += int(final[u] != hyp_ar)
    assert n_ctc > 0


# ---------------------------------------------------------------------------
# next-character prior and the training regulariser
# ---------------------------------------------------------------------------
def _prior_golden():
    import json
    from pathlib import Path
    return json.loads((Path(__file__).parent / "golden" / "ns_prior_golden.json").read_text(encoding="utf-8"))


def _f32(b64, shape):
    import base64
    return np.frombuffer(base64.b64decode(b64), np.float32).reshape(shape)


@pytest.mark.gpu
def test_cond_dist_matches_reference_values():
    """nsd_cond_dist against CharNgramLM.cond_dist values recorded from both reference copies.
    float32; the 37-term sum is taken in another order than torch's, hence 1e-6 relative."""
    from silent_speech_amd.prior import cond_dist_batch, tables_from_lm
    from silent_speech_amd.tables import NgramCounts
    n = 0
    for rec in _prior_golden()["cond_dist"]:
        words = synth.make_lexicon(rec["n_words"], rec["lex_seed"], max_len=6)
        corpus = synth.make_sentences(words, 300, rec["corpus_seed"])[:rec["n_corpus"]]
        tab = tables_from_lm(NgramCounts(rec["order"]).fit(corpus))
        got = cond_dist_batch(tab, rec["contexts"], form=rec["form"]).cpu().numpy()
        want = _f32(rec["probs_f32"], (len(rec["contexts"]), 37))
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0, err_msg=f"{rec['form']} order {rec['order']}")
        n += len(want)
    assert n > 150
    # a character outside the alphabet matches nothing, as in the reference's dict lookups
    lm = O.NgramLM(3).fit(corpus)
    tab = tables_from_lm(NgramCounts(3).fit(corpus))
    got = cond_dist_batch(tab, ["a!", "!a", ""], form="U").cpu().numpy()
    for row, c in zip(got, ["a!", "!a", ""]):
        np.testing.assert_allclose(row, O.cond_dist(lm, c, synth.BASE_CHARS, "U"), rtol=1e-6)


@pytest.mark.gpu
def test_lm_regularizer_matches_reference_values():
    """Loss and gradient of the regulariser against values recorded from train_emg_llm.py's own function
    (torch autograd), and against the CPU restatement on a larger batch.  float32 end to end: loss within
    2e-6 relative, gradient within 2e-5 relative."""
    from silent_speech_amd.prior import ns_lm_regularizer, tables_from_lm
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    for rec in _prior_golden()["regularizer"]:
        words = synth.make_lexicon(rec["n_words"], rec["lex_seed"], max_len=6)
        corpus = synth.make_sentences(words, 300, rec["corpus_seed"])[:rec["n_corpus"]]
        tab = tables_from_lm(NgramCounts(rec["order"]).fit(corpus))
        rs = np.random.RandomState(rec["seed"])
        logits = (rs.standard_normal((rec["B"], rec["L"], 40)) * rec["scale"]).astype(np.float32)
        x = torch.from_numpy(logits).to(dev).requires_grad_(True)
        out_ids = torch.tensor(rec["out_ids"], dtype=torch.int64, device=dev)
        loss = ns_lm_regularizer(x, None, out_ids, tab)
        want = float(_f32(rec["loss_f32"], (1,))[0])
        assert abs(float(loss.detach()) - want) <= 2e-6 * max(1.0, abs(want)), rec["name"]
        (loss * 1.0).backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), _f32(rec["grad_f32"], logits.shape), rtol=2e-5, atol=2e-8,
                                   err_msg=rec["name"])
    # a training-sized batch against the restatement, with an upstream factor as in the total loss
    words = synth.make_lexicon(200, 0)
    corpus = synth.make_sentences(words, 2000, 1)
    lm = O.NgramLM(3).fit(corpus)
    tab = tables_from_lm(NgramCounts(3).fit(corpus))
    B, L = 16, 72
    rs = np.random.RandomState(7)
    logits = (rs.standard_normal((B, L, 40)) * 3).astype(np.float32)
    out_ids = np.full((B, L), -100, np.int64)
    for i, s in enumerate(synth.make_sentences(words, B, 55, 3, 7)):
        toks = synth.text_to_tokens(s)[:L - 1] + [synth.EOS_ID]
        out_ids[i, :len(toks)] = toks
    x = torch.from_numpy(logits).to(dev).requires_grad_(True)
    loss = ns_lm_regularizer(x, None, torch.from_numpy(out_ids).to(dev), tab)
    (0.05 * loss).backward()
    wl, wg = O.lm_regularizer(lm, logits, out_ids)
    assert abs(float(loss.detach()) - float(wl)) <= 2e-6 * abs(float(wl))
    np.testing.assert_allclose(x.grad.cpu().numpy(), 0.05 * wg, rtol=2e-5, atol=2e-8)
    # same bits on a second call (fixed-order reduction)
    loss2 = ns_lm_regularizer(x.detach(), None, torch.from_numpy(out_ids).to(dev), tab)
    assert float(loss2) == float(loss.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["I", "U"])
def test_full_size_batch_properties(variant):
    """The benchmark's full batch (4096 utterances, beam 16, 5-gram, 64 steps), checked through properties
    that do not need the CPU checker on every utterance: a second decode gives the same bits; a slice of the
    batch decoded on its own gives the same ids / lengths / fp64 scores as inside the batch (utterances are
    independent); the lanes-per-utterance knob changes nothing; and a 48-utterance sample equals the C checker."""
    from silent_speech_amd import synthetic as S
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    N, B, T, seed, margin = 4096, 16, 64, 20260403, 60.0
    words = S.make_lexicon(200, 0)
    corpus = S.make_sentences(words, 20000, 1)
    targets = S.make_sentences(words, N, 1000, 3, 7)
    tg, tl = S.pack_targets(targets)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    floor = variant == "U"
    tables = DeviceTables.from_words(words, NgramCounts(5, floor_vsz=floor).fit(corpus))

    def run(lo, hi, lanes=0):
        dec = NSDecoder(tables, B, T, variant=variant, lm_weight=0.05, oov_penalty=1.0, lanes_per_utt=lanes)
        ids, lens, sc = dec.decode_synth_native(tgd[lo:hi].contiguous(), tld[lo:hi].contiguous(), base_seed=seed,
                                                margin=margin, utt0=lo)
        torch.cuda.synchronize()
        return ids.cpu().numpy().copy(), lens.cpu().numpy().copy(), sc.cpu().numpy().copy()

    ids, lens, sc = run(0, N)
    ids2, lens2, sc2 = run(0, N)
    assert (ids == ids2).all() and (lens == lens2).all() and (sc.view(np.int64) == sc2.view(np.int64)).all()
    for lo, hi, lanes in ((1000, 1064, 0), (4000, 4096, 0), (0, N, 32), (2048, 2560, 16)):
        i3, l3, s3 = run(lo, hi, lanes)
        assert (l3 == lens[lo:hi]).all(), (lo, hi, lanes)
        assert (i3 == ids[lo:hi]).all(), (lo, hi, lanes)
        assert (s3.view(np.int64) == sc[lo:hi].view(np.int64)).all(), (lo, hi, lanes)
    # a sample against the CPU checker (which numbers utterances from 0: take the first ones)
    n = 48
    ref = cbind.Model(words, corpus, 5, floor).decode_synth(
        variant=variant, beam=B, max_new_tokens=T, beta=0.05, gamma=1.0, kappa=0.0, margin=margin, base_seed=seed,
        targets=tg[:n], target_lens=tl[:n], trace=False)
    assert (ref["lens"] == lens[:n]).all()
    for u in range(n):
        assert (ref["ids"][u, :lens[u]] == ids[u, :lens[u]]).all(), u
        assert float(ref["scores"][u]).hex() == float(sc[u]).hex(), u


@pytest.mark.gpu
def test_two_utterances_per_wave_overflowing_at_different_times(dev):
    """Beam 32 on 32 lanes per utterance, flat logits, a three-word lexicon with unspellable and multi-word
    entries: nearly every hypothesis is in the top-k fall-back, the candidate buffer overflows several times per
    step, and the two utterances that share a wave reach their overflows at different stages (one of them may
    already be fully scored when the other one forces the scoring pass)."""
    case = dict(variant="E", beam=32, lm_order=2, beta=0.0, margin=0.0, gamma=1.0, kappa=0.35, alpha=0.0,
                min_len_for_eos=8, max_new_tokens=24, n_words=3, lex_seed=11, max_len=3, n_corpus=2000, corpus_seed=27,
                n_utts=200, target_seed=79, base_seed=314692659, min_words=1, max_words=1, lanes=32, name="overflow32",
                allow_eos_anytime=False, alphabet=synth.BASE_CHARS[:36], extra_words=["zq'x", "ice cream cone", "a b"])
    ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
    assert err is None
    _compare_with_trace(ws, ids, lens, scores, ref, case["name"])


@pytest.mark.gpu
def test_one_gpu_share_of_the_million_utterance_batch():
    """One GPU's share of the 1 000 000-utterance configuration over 8 GPUs (131 072 utterances, beam 16; the
    launcher then puts two utterances on a wave): slices of the batch decoded on their own (one utterance per
    wave) give the same ids / lengths / fp64 scores, a second decode gives the same bits, and the first
    utterances equal the C checker."""
    from silent_speech_amd import synthetic as S
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    N, B, T, seed, margin = 131072, 16, 64, 20260403, 60.0
    words = S.make_lexicon(200, 0)
    corpus = S.make_sentences(words, 20000, 1)
    tg, tl = S.pack_targets(S.make_sentences(words, N, 1000, 3, 7))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    tables = DeviceTables.from_words(words, NgramCounts(5, floor_vsz=False).fit(corpus))
    dec = NSDecoder(tables, B, T, variant="I", lm_weight=0.05, oov_penalty=1.0)

    def run(lo, hi):
        ids, lens, sc = dec.decode_synth_native(tgd[lo:hi].contiguous(), tld[lo:hi].contiguous(), base_seed=seed,
                                                margin=margin, utt0=lo)
        torch.cuda.synchronize()
        return ids.cpu().numpy().copy(), lens.cpu().numpy().copy(), sc.cpu().numpy().copy()

    ids, lens, sc = run(0, N)
    ids2, lens2, sc2 = run(0, N)
    assert (ids == ids2).all() and (lens == lens2).all() and (sc.view(np.int64) == sc2.view(np.int64)).all()
    for lo, hi in ((0, 512), (65000, 66000), (N - 300, N)):
        i3, l3, s3 = run(lo, hi)
        assert (l3 == lens[lo:hi]).all(), (lo, hi)
        assert (i3 == ids[lo:hi]).all(), (lo, hi)
        assert (s3.view(np.int64) == sc[lo:hi].view(np.int64)).all(), (lo, hi)
    n = 24
    ref = cbind.Model(words, corpus, 5, False).decode_synth(
        variant="I", beam=B, max_new_tokens=T, beta=0.05, gamma=1.0, kappa=0.0, margin=margin, base_seed=seed,
        targets=tg[:n], target_lens=tl[:n], trace=False)
    assert (ref["lens"] == lens[:n]).all()
    for u in range(n):
        assert (ref["ids"][u, :lens[u]] == ids[u, :lens[u]]).all(), u
        assert float(ref["scores"][u]).hex() == float(sc[u]).hex(), u


@pytest.mark.gpu
def test_step_into_a_second_state_buffer_equals_in_place():
    """nsd_step may write the new hypothesis state into other arrays than it reads (double buffering): the
    result equals stepping in place, including utterances that are already finished (their state is copied)."""
    import ctypes as C
    from silent_speech_amd import lib
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, Workspace
    dev = torch.device("cuda", 0)
    words = synth.make_lexicon(200, 0)
    corpus = synth.make_sentences(words, 2000, 1)
    targets = synth.make_sentences(words, 96, 9, 1, 3)          # short targets: variant U finishes early
    tg, tl = _pack_targets(targets)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    tables = DeviceTables.from_words(words, O.NgramLM(5).fit(corpus))
    N, B, T = len(targets), 8, 40
    dec = NSDecoder(tables, B, T, variant="U", lm_weight=0.05)
    ids_a, lens_a, sc_a = dec.decode_synth(tgd, tld, base_seed=11, margin=30.0, check_every=0)
    ids_a, lens_a, sc_a = ids_a.cpu().numpy().copy(), lens_a.cpu().numpy().copy(), sc_a.cpu().numpy().copy()

    L = lib.load()
    w = [Workspace(N, B, T, dev), Workspace(N, B, T, dev)]
    # a ping-pong pair shares one done counter (every step adds to the word of its output state)
    w[1].done_count = w[0].done_count
    w[1].ptrs.done_count = w[0].done_count.data_ptr()
    dec.init(w[0])
    st = torch.cuda.current_stream().cuda_stream
    for t in range(T):
        src, dst = w[t & 1], w[1 - (t & 1)]
        lib.check(L.nsd_synth_logits(C.byref(src.ptrs), 11, 0, tgd.data_ptr(), tld.data_ptr(), tgd.shape[1], 30.0,
                                     src.logits.data_ptr(), N, B, st), "nsd_synth_logits")
        # back-pointers and tokens of all steps are kept in the first workspace
        lib.check(L.nsd_step(tables.handle, C.byref(dec.params), src.logits.data_ptr(), C.byref(src.ptrs),
                             C.byref(dst.ptrs), w[0].backptr.data_ptr() + t * N * B, w[0].token.data_ptr() + t * N * B,
                             None, N, B, 40, st), "nsd_step")
    last = w[T & 1]
    out_ids = torch.zeros(N, T + 1, dtype=torch.uint8, device=dev)
    out_len = torch.zeros(N, dtype=torch.int32, device=dev)
    out_sc = torch.zeros(N, dtype=torch.float64, device=dev)
    lib.check(L.nsd_backtrace(C.byref(last.ptrs), w[0].backptr.data_ptr(), w[0].token.data_ptr(), T, N, B,
                              out_ids.data_ptr(), out_len.data_ptr(), out_sc.data_ptr(), st), "nsd_backtrace")
    torch.cuda.synchronize()
    assert (out_len.cpu().numpy() == lens_a).all()
    assert (out_ids.cpu().numpy() == ids_a).all()
    assert (out_sc.cpu().numpy().view(np.int64) == sc_a.view(np.int64)).all()
    assert bool((last.utt[:N] & lib.U_DONE).all())              # every utterance finished well before step 40
    assert int(w[0].done_count.item()) == N


@pytest.mark.gpu
@pytest.mark.parametrize("beam,lanes,n_utts", [(60, 0, 48), (28, 32, 96), (15, 16, 128)])
def test_largest_lds_footprints_of_the_length_normalised_copy(dev, beam, lanes, n_utts):
    """Beams just below a lane-group width with the copy of the decoder that also keeps raw scores in LDS
    (eval_ar.py): the largest per-workgroup LDS footprints the launcher can ask for."""
    case = dict(variant="E", beam=beam, lm_order=3, beta=0.1, margin=6.0, gamma=1.0, kappa=0.35, alpha=0.6,
                min_len_for_eos=4, max_new_tokens=20, n_words=300, lex_seed=5, max_len=6, n_corpus=2000, corpus_seed=6,
                n_utts=n_utts, target_seed=7, base_seed=1234, min_words=2, max_words=4, lanes=lanes,
                name=f"lds_E_{beam}_{lanes}")
    # a plain lexicon never empties a hypothesis' options, so the top-k fall-back (which raises for a beam wider
    # than the vocabulary in this copy) is not reached
    ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
    assert err is None
    _compare_with_trace(ws, ids, lens, scores, ref, case["name"])


@pytest.mark.gpu
@pytest.mark.parametrize("variant,beam,lanes", [("I", 16, 0), ("U", 32, 0), ("E", 8, 16), ("I", 16, 32)])
def test_trie_levels_staged_in_lds_change_nothing(dev, variant, beam, lanes):
    """The upper trie levels copied to LDS by every workgroup (a tuning knob of the launcher): same parents, tokens,
    masks and fp64 scores as with every node record read from memory, on a lexicon deeper than what is staged."""
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    words = synth.make_lexicon(4000, 3, synth.BASE_CHARS[:36], 2, 9)
    corpus = synth.make_sentences(words, 2000, 4)
    tg, tl = _pack_targets(synth.make_sentences(words, 300, 5, 2, 4))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    tables = DeviceTables.from_words(words, O.NgramLM(5, floor_vsz=(variant == "U")).fit(corpus))
    outs = []
    for kn in (0, 100, 1024, 5000):
        dec = NSDecoder(tables, beam, 30, variant=variant, lm_weight=0.05, word_bonus=0.35, min_len_for_eos=4,
                        lanes_per_utt=lanes, lds_trie_nodes=kn)
        ws = dec.workspace(300, with_masks=True)
        ids, lens, sc = dec.decode_synth(tgd, tld, base_seed=8, margin=7.0, ws=ws, check_every=0)
        torch.cuda.synchronize()
        outs.append([x.cpu().numpy().copy() for x in (ids, lens, sc, ws.backptr, ws.token, ws.masks)])
    for o in outs[1:]:
        assert (o[1] == outs[0][1]).all() and (o[0] == outs[0][0]).all()
        assert (o[2].view(np.int64) == outs[0][2].view(np.int64)).all()
        assert (o[5] == outs[0][5]).all()
    ref = cbind.Model(words, corpus, 5, variant == "U").decode_synth(
        variant=variant, beam=beam, max_new_tokens=30, beta=0.05, gamma=1.0, kappa=0.35, min_len_for_eos=4, margin=7.0,
        base_seed=8, targets=tg, target_lens=tl, trace=False)
    assert (ref["lens"] == outs[2][1]).all()
    for u in range(300):
        assert (ref["ids"][u, :ref["lens"][u]] == outs[2][0][u, :ref["lens"][u]]).all(), u
