"""GPU parity against outputs recorded from the reference itself at benchmark sizes
(tests/golden/ns_golden_tail.json, written by tests/golden/make_golden_tail.py): decoded ids and transcripts
bit-exact, fp64 scores within 1e-5 relative (the tolerance that covers the library's own deterministic fp32
log-softmax against torch's), near-ties at the beam cut listed; the CTC proposal and both rerank rules; a decode
from a saved table file."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle import ns_oracle as O
from tests.golden_util import case_corpus, case_targets, case_words, ctc_frames, target_tokens

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5            # score tolerance against the reference (fp32 log-softmax rounding)
NEAR_TIE = 2e-5           # a beam cut closer than this may legitimately fall the other way


def _pack(targets):
    toks = [target_tokens(t) for t in targets]
    S = max(1, max(len(t) for t in toks))
    tg = np.zeros((len(toks), S), np.int8)
    for i, t in enumerate(toks):
        tg[i, :len(t)] = t
    return tg, np.array([len(t) for t in toks], np.int32)


@pytest.mark.parametrize("idx", range(21))
def test_reference_ids_and_scores_at_benchmark_sizes(golden_tail, idx):
    from silent_speech_amd import TextTransform
    from silent_speech_amd.decoder import DeviceTables, NSDecoder, ids_to_text
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    case = golden_tail["scored"][idx]
    words = case_words(case)
    corpus = case_corpus(case, words)
    targets = case_targets(case, words)
    tg, tl = _pack(targets)
    lm = NgramCounts(case["lm_order"], floor_vsz=(case["variant"] == "U")).fit(corpus)
    dec = NSDecoder(DeviceTables.from_words(words, lm), case["beam"], case["max_new_tokens"], variant=case["variant"],
                    lm_weight=case["beta"], oov_penalty=case["gamma"], word_bonus=case["kappa"],
                    length_penalty_alpha=case["alpha"], min_len_for_eos=case["min_len_for_eos"])
    ids, lens, scores = dec.decode_synth_native(torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev),
                                                base_seed=case["base_seed"], margin=case["margin"])
    torch.cuda.synchronize()
    ids, lens, scores = ids.cpu().numpy(), lens.cpu().numpy(), scores.cpu().numpy()
    near, differ = [], []
    worst = 0.0
    for u, rec in enumerate(case["utts"]):
        got = ids[u, :lens[u]].tolist()
        tie = rec["min_cut_gap"] is not None and rec["min_cut_gap"] < NEAR_TIE
        if tie:
            near.append((u, rec["min_cut_gap"]))
        if got != rec["ids"]:
            differ.append(u)
            assert tie, (case["name"], u, "ids differ from the reference although no beam cut was close")
            continue
        assert ids_to_text(TextTransform, got) == rec["text"]
        want = float.fromhex(rec["score"])
        rel = abs(float(scores[u]) - want) / max(1.0, abs(want))
        worst = max(worst, rel)
        assert rel <= REL_TOL, (case["name"], u, float(scores[u]), want)
    print(f"{case['name']}: worst relative score error {worst:.2e}; near-ties at the beam cut (< {NEAR_TIE}): {near}; "
          f"of those decoded differently: {differ}")
    assert len(differ) <= len(near)


@pytest.mark.parametrize("which", ["I", "E"])
def test_ctc_proposal_and_rerank_against_the_drivers_outputs(golden_tail, which):
    """nsd_ctc_greedy, the LM choice (inference_emg_llm.py:441-464) and the joint choice (eval_ar.py:554-596) on the
    inputs the reference's own statements were run on."""
    from silent_speech_amd import TextTransform
    from silent_speech_amd.decoder import (DeviceTables, ctc_greedy_batch, transcribe_batch, transcribe_batch_joint)
    from silent_speech_amd.tables import NgramCounts
    dev = torch.device("cuda", 0)
    doc = golden_tail["ctc_tail_" + which]
    words = synth.make_lexicon(doc["n_words"], doc["lex_seed"], max_len=doc["max_len"])
    corpus = synth.make_sentences(words, doc["n_corpus"], doc["corpus_seed"])
    tables = DeviceTables.from_words(words, NgramCounts(doc["order"], floor_vsz=False).fit(corpus))
    trie = O.build_trie(words)
    rows = doc["rows"]
    N = len(rows)
    Tp = max(r["Tp"] for r in rows)
    x = np.zeros((N, Tp, 40), np.float32)
    x[:, :, 39] = 9.0                                                     # blank frames behind the shorter ones
    for u, r in enumerate(rows):
        x[u, :r["Tp"]] = ctc_frames(r["ctc_seed"], r["ctc_target"], r["Tp"], wrong=r["wrong"])
    xd = torch.from_numpy(x).to(dev)
    ci, cl = ctc_greedy_batch(xd)
    ci, cl = ci.cpu().numpy(), cl.cpu().numpy()
    for u, r in enumerate(rows):
        assert ci[u, :cl[u]].tolist() == r["ctc_ids"], u
    W = max(len(r["hyp_ar_raw"]) for r in rows) + 2
    ar = np.full((N, W), 39, np.uint8)
    al = np.zeros(N, np.int32)
    for u, r in enumerate(rows):
        seq = [37] + [synth.BASE_CHARS.index(c) for c in r["hyp_ar_raw"]] + [38]
        ar[u, :len(seq)] = seq
        al[u] = len(seq)
    ar_d, al_d = torch.from_numpy(ar).to(dev), torch.from_numpy(al).to(dev)
    if which == "I":
        got = transcribe_batch(TextTransform, ar_d, al_d, tables, trie, words, ctc_logits=xd)
        assert got == [r["hyp"] for r in rows]
        return
    tgt_tokens = [target_tokens(r["target"]) for r in rows]

    def teacher_logits(seq, lens):
        sq, ln = seq.cpu().numpy(), lens.cpu().numpy()
        out = np.zeros((sq.shape[0], sq.shape[1] - 1, 40), np.float32)
        for m in range(sq.shape[0]):
            u = m % N
            for i in range(1, ln[m]):
                out[m, i - 1] = synth.logits_row(synth.utt_seed(doc["step_seed"], u), sq[m, :i].tolist(),
                                                 tgt_tokens[u], doc["step_margin"])
        return torch.from_numpy(out).to(seq.device)

    got, scores = transcribe_batch_joint(TextTransform, ar_d, al_d, tables, trie, words, xd, teacher_logits, doc["beta"])
    for u, r in enumerate(rows):
        want = [float.fromhex(h) for h in r["scores"]]
        for c in range(2):
            assert abs(scores[u, c] - want[c]) <= REL_TOL * max(1.0, abs(want[c])), (u, c, scores[u, c], want[c])
        if abs(want[0] - want[1]) > 1e-3:
            assert got[u] == r["hyp"], u


def test_decode_from_a_saved_table_file_equals_the_direct_path(tmp_path):
    """save_tables -> load_tables -> DeviceTables: the decode reads the same bytes as from freshly compiled tables."""
    from silent_speech_amd import synthetic as S
    from silent_speech_amd import tables as T
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    dev = torch.device("cuda", 0)
    words = S.make_lexicon(3000, 7, S.BASE_CHARS[:36], 2, 9) + ["new york", "don't"]
    plain = words[:3000]
    lm = T.NgramCounts(5, floor_vsz=False).fit(S.make_sentences(plain, 3000, 8))
    nodes = T.flatten_trie(T.HostTrie.from_words(words))
    ng = T.compile_ngram(lm)
    path = tmp_path / "tables.npz"
    T.save_tables(str(path), nodes, ng)
    n2, g2 = T.load_tables(str(path))
    tg, tl = S.pack_targets(S.make_sentences(plain, 300, 9, 2, 5))
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    outs = []
    for tab in (DeviceTables(nodes, ng), DeviceTables(n2, g2)):
        for variant in ("I", "E"):
            dec = NSDecoder(tab, 16, 40, variant=variant, lm_weight=0.1, oov_penalty=1.0, word_bonus=0.35,
                            min_len_for_eos=4)
            ids, lens, sc = dec.decode_synth_native(tgd, tld, base_seed=5, margin=9.0)
            torch.cuda.synchronize()
            outs.append((ids.cpu().numpy().copy(), lens.cpu().numpy().copy(), sc.cpu().numpy().copy()))
    for a, b in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert (a[1] == b[1]).all() and (a[0] == b[0]).all() and (a[2].view(np.int64) == b[2].view(np.int64)).all()
    assert len({bytes(r) for r in outs[0][0]}) > 200
