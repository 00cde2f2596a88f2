"""The restated next-character prior and training regulariser against values recorded from the reference
(tests/golden/ns_prior_golden.json, written by tests/golden/make_golden_prior.py)."""
import base64
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import ns_oracle as O
from oracle import synth

GOLD = Path(__file__).parent / "golden" / "ns_prior_golden.json"


@pytest.fixture(scope="module")
def prior_golden():
    return json.loads(GOLD.read_text(encoding="utf-8"))


def f32(b64, shape=None):
    a = np.frombuffer(base64.b64decode(b64), np.float32)
    return a.reshape(shape) if shape is not None else a


def corpus_of(rec):
    words = synth.make_lexicon(rec["n_words"], rec["lex_seed"], max_len=6)
    return words, synth.make_sentences(words, 300, rec["corpus_seed"])[:rec["n_corpus"]]


def regularizer_inputs(rec):
    rs = np.random.RandomState(rec["seed"])
    logits = (rs.standard_normal((rec["B"], rec["L"], 40)) * rec["scale"]).astype(np.float32)
    return logits, np.array(rec["out_ids"], np.int64)


def test_cond_dist_matches_reference(prior_golden):
    n = 0
    for rec in prior_golden["cond_dist"]:
        _, corpus = corpus_of(rec)
        lm = O.NgramLM(rec["order"]).fit(corpus)
        want = f32(rec["probs_f32"], (len(rec["contexts"]), 37))
        for c, w in zip(rec["contexts"], want):
            got = O.cond_dist(lm, c, synth.BASE_CHARS, form=rec["form"])
            # float32 throughout; only the order of the 37-term sum differs from torch's
            np.testing.assert_allclose(got, w, rtol=1e-6, atol=0)
            n += 1
    assert n > 150


def test_regularizer_matches_reference(prior_golden):
    for rec in prior_golden["regularizer"]:
        _, corpus = corpus_of(rec)
        lm = O.NgramLM(rec["order"]).fit(corpus)
        logits, out_ids = regularizer_inputs(rec)
        loss, grad = O.lm_regularizer(lm, logits, out_ids)
        want = float(f32(rec["loss_f32"])[0])
        assert abs(float(loss) - want) <= 2e-6 * max(1.0, abs(want)), rec["name"]
        np.testing.assert_allclose(grad, f32(rec["grad_f32"], logits.shape), rtol=2e-5, atol=2e-8, err_msg=rec["name"])
