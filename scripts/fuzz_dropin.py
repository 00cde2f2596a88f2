"""Randomised sweep of the reference-signature entry point: constrained_beam_decode(step_fn, TT, ...) of the
package against the Python restatement, utterance by utterance: same sequence of step_fn calls (the live
hypotheses in rank order at every step), same ids, same error behaviour.  usage: fuzz_dropin.py [n] [seed]"""
import random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from oracle import synth
from oracle import ns_oracle as O
from silent_speech_amd import TextTransform
from silent_speech_amd.decoder import constrained_beam_decode
from tests.golden_util import case_corpus, case_targets, case_words, decode_kwargs, target_tokens

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
for k in range(n_cases):
    variant = rng.choice("UIE")
    case = dict(variant=variant, beam=rng.choice([1, 2, 4, 8, 16, 33, 40, 41, 64]), lm_order=rng.choice([0, 1, 2, 3, 5]),
                beta=rng.choice([0.0, 0.05, 0.3]), margin=rng.choice([0.0, 3.0, 10.0, 60.0]), gamma=rng.choice([0.0, 1.0]),
                kappa=0.35, alpha=rng.choice([0.0, 0.6]), min_len_for_eos=rng.choice([0, 8]),
                max_new_tokens=rng.choice([0, 1, 6, 20]), n_words=rng.choice([3, 50, 400]), lex_seed=rng.randrange(100),
                max_len=rng.choice([3, 8]), n_corpus=rng.choice([0, 40, 500]), corpus_seed=rng.randrange(100), n_utts=2,
                target_seed=rng.randrange(100), base_seed=rng.randrange(1 << 30), min_words=1, max_words=3)
    if rng.random() < 0.3:
        case["extra_words"] = rng.sample(["don't", "it's", "new york", "a b", "x-ray"], 2)
    quant = rng.choice([0, 0, 1, 2])
    if quant:
        case["beta"] = rng.choice([0.0, case["beta"]])
    words = case_words(case)
    corpus = case_corpus(case, words)
    trie = O.build_trie(words)
    lm = O.NgramLM(case["lm_order"], floor_vsz=(variant == "U")).fit(corpus) if case["lm_order"] > 0 else None
    kw = decode_kwargs(case)
    if variant == "U":
        kw["allow_eos_anytime"] = rng.random() < 0.8
    for u, target in enumerate(case_targets(case, words)):
        useed = synth.utt_seed(case["base_seed"], u)
        tgt = target_tokens(target)
        calls = [[], []]
        def make(which):
            def step_fn(ids):
                calls[which].append(list(ids))
                row = synth.logits_row(useed, ids, tgt, case["margin"])
                if quant:                                   # coarse logits: many exactly equal scores
                    row = (np.round(row * quant) / quant).astype(np.float32)
                return torch.from_numpy(row).unsqueeze(0)
            return step_fn
        out = []
        for which, fn in enumerate((lambda sf: O.constrained_beam_decode(sf, O.TextTransformLike, case["beam"], case["max_new_tokens"], trie, lm,
                                                                          log_softmax=O.lsm_det, **kw),
                                    lambda sf: constrained_beam_decode(sf, TextTransform, case["beam"], case["max_new_tokens"], trie, lm, **kw))):
            try:
                out.append(("ids", fn(make(which))))
            except (RuntimeError, ValueError) as e:
                out.append(("raises", type(e).__name__))
        assert out[0] == out[1], (case, u, out)
        assert calls[0] == calls[1], (case, u, "step_fn call sequences differ")
    print(f"{k:3d} ok  {variant} beam {case['beam']} order {case['lm_order']} T {case['max_new_tokens']} -> {out[1][0]}", flush=True)
print("all", n_cases, "cases equal")
