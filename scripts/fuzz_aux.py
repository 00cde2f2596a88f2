"""Randomised sweep of the auxiliary kernels against the CPU restatements: nsd_lm_logp (bit-exact), nsd_rerank_lm
(choice and scores), nsd_ctc_greedy (ids), nsd_rerank_gather (bit-exact), nsd_cond_dist and the LM regulariser
(float32 tolerances of the tests).  usage: fuzz_aux.py [n_rounds] [seed]"""
import random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from oracle import cbind, synth
from oracle import ns_oracle as O
from silent_speech_amd.decoder import (DeviceTables, ctc_greedy_batch, lm_logp_batch, rerank_gather_batch,
                                       rerank_lm_batch)
from silent_speech_amd.prior import cond_dist_batch, ns_lm_regularizer, tables_from_lm

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda", 0)
ALPHA = synth.BASE_CHARS


def rand_text(maxlen):
    n = rng.randrange(0, maxlen + 1)
    return "".join(rng.choice(ALPHA if rng.random() < 0.8 else " ") for _ in range(n))


def pack(texts, width=None):
    S = max(1, width or max(len(t) for t in texts))
    ids = np.zeros((len(texts), S), np.uint8)
    lens = np.array([len(t) for t in texts], np.int32)
    for i, t in enumerate(texts):
        ids[i, :len(t)] = synth.text_to_tokens(t)
    return torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)


for k in range(rounds):
    order = rng.choice([1, 1, 2, 3, 5, 7, 10])
    n_words, n_corpus = rng.choice([3, 60, 400]), rng.choice([0, 5, 300])
    floor = rng.random() < 0.5
    words = synth.make_lexicon(n_words, rng.randrange(100), max_len=rng.choice([3, 8]))
    corpus = synth.make_sentences(words, n_corpus, rng.randrange(100)) if n_corpus else []
    lm = O.NgramLM(order, floor_vsz=floor).fit(corpus)
    tables = DeviceTables.from_words(words, lm, floor_vsz=floor)
    # ---- whole-string LM score
    texts = [rand_text(30) for _ in range(150)] + corpus[:20] + ["", " ", "  a  "]
    got = lm_logp_batch(tables, *pack(texts)).cpu().numpy()
    for i, t in enumerate(texts):
        assert float(got[i]).hex() == lm.logp(t).hex(), ("lm_logp", order, n_corpus, repr(t))
    # ---- LM rerank of an AR hypothesis (BOS ... EOS, specials inside) against a CTC proposal
    n = 120
    ar, ctc = [], []
    for _ in range(n):
        body = [rng.randrange(0, 37) if rng.random() < 0.9 else rng.choice([37, 39]) for _ in range(rng.randrange(0, 25))]
        ar.append([37] + body + ([38] if rng.random() < 0.7 else []) + [rng.randrange(0, 40) for _ in range(rng.randrange(0, 3))])
        ctc.append([rng.randrange(0, 37) for _ in range(rng.randrange(0, 25))])
    def as_tensor(seqs):
        S = max(1, max(len(s) for s in seqs))
        a = np.zeros((len(seqs), S), np.uint8); l = np.array([len(s) for s in seqs], np.int32)
        for i, s in enumerate(seqs):
            a[i, :len(s)] = s
        return torch.from_numpy(a).to(dev), torch.from_numpy(l).to(dev)
    choice, sc = rerank_lm_batch(tables, *as_tensor(ar), *as_tensor(ctc))
    choice, sc = choice.cpu().numpy(), sc.cpu().numpy()
    for i in range(n):
        ta = O.ids_to_text(O.TextTransformLike, ar[i])
        tc = O.ids_to_text(O.TextTransformLike, [37] + ctc[i] + [38])
        fa, fc = lm.logp(ta) - 0.002 * len(ta), lm.logp(tc) - 0.002 * len(tc)
        assert float(sc[i, 0]).hex() == float(fa).hex() and float(sc[i, 1]).hex() == float(fc).hex(), ("rerank score", i, ta, tc)
        assert int(choice[i]) == (1 if fc > fa else 0), ("rerank choice", i)
    # ---- CTC greedy
    N, Tp = rng.choice([1, 9, 40]), rng.choice([1, 17, 64, 200])
    x = np.random.RandomState(rng.randrange(1 << 30)).randn(N, Tp, 40).astype(np.float32)
    if rng.random() < 0.5:
        x = np.round(x)                                        # many exact ties: the first index wins
    ids, lens = ctc_greedy_batch(torch.from_numpy(x).to(dev))
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    for u in range(N):
        assert list(ids[u, :lens[u]]) == O.ctc_greedy(x[u], O.TextTransformLike), ("ctc", u)
    # ---- teacher-forced gather
    N, Lm = rng.choice([1, 20]), rng.choice([1, 13, 70])
    rs = np.random.RandomState(rng.randrange(1 << 30))
    lg = (rs.randn(N, Lm, 40) * rng.choice([0.1, 3.0, 30.0])).astype(np.float32)
    tgt = rs.randint(0, 40, (N, Lm)).astype(np.uint8)
    ln = rs.randint(0, Lm + 1, N).astype(np.int32)
    got = rerank_gather_batch(torch.from_numpy(lg).to(dev), torch.from_numpy(tgt).to(dev), torch.from_numpy(ln).to(dev)).cpu().numpy()
    lsm = cbind.log_softmax_rows(lg)
    for u in range(N):
        tot = 0.0
        for i in range(ln[u]):
            tot += float(lsm[u, i, tgt[u, i]])
        assert float(got[u]).hex() == tot.hex(), ("gather", u)
    # ---- next-character prior (both forms) and the regulariser: max(1, |V|) as both cond_dist copies use
    lm1 = O.NgramLM(order, floor_vsz=True).fit(corpus)
    tp = tables_from_lm(lm1)
    ctxs = [rand_text(12) for _ in range(60)] + ["", "a", "A!b"]
    for form in ("U", "T"):
        got = cond_dist_batch(tp, ctxs, form=form).cpu().numpy()
        for c, row in zip(ctxs, got):
            np.testing.assert_allclose(row, O.cond_dist(lm1, c, synth.BASE_CHARS, form), rtol=1e-6, atol=0,
                                       err_msg=f"cond_dist {form} order {order} {c!r}")
    Bq, Lq = rng.choice([1, 5]), rng.choice([1, 9, 30])
    rs = np.random.RandomState(rng.randrange(1 << 30))
    logits = (rs.randn(Bq, Lq, 40) * 2).astype(np.float32)
    out_ids = rs.randint(0, 40, (Bq, Lq)).astype(np.int64)
    out_ids[rs.rand(Bq, Lq) < 0.3] = -100
    xg = torch.from_numpy(logits).to(dev).requires_grad_(True)
    loss = ns_lm_regularizer(xg, None, torch.from_numpy(out_ids).to(dev), tp)
    wl, wg = O.lm_regularizer(lm1, logits, out_ids)
    if loss.requires_grad:
        loss.backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), wg, rtol=3e-5, atol=3e-8, err_msg="regulariser gradient")
    assert abs(float(loss.detach()) - float(wl)) <= 3e-6 * max(1.0, abs(float(wl))), ("regulariser", float(loss.detach()), float(wl))
    print(f"{k:3d} ok  order {order} words {n_words} corpus {n_corpus} floor {floor}", flush=True)
print("all", rounds, "rounds equal")
