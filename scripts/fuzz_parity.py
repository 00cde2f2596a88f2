"""Randomised parity sweep: the HIP decoder against the C checker (parents, tokens, lexicon masks per step,
fp64 scores, ids) over random decoder variants, beams, n-gram orders, weights, lexicon sizes, logit margins
and lanes-per-utterance.  usage: fuzz_parity.py [n_cases] [seed] [only_case] [hot]
With `hot` the cases are drawn around the shapes that have nsd_step instantiations of their own (beams 8 / 16 / 32
with a 5-gram model, 32 or 64 lanes per utterance), with negative LM weights and with lexicons large enough that the
bound sweep runs."""
import random, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from oracle import synth
from tests.test_gpu_parity import _compare_with_trace, _run_case_batched



def random_cases(n_cases, seed):
    """The sweep's cases, in order (the stream of one seeded generator: case k depends on the k before it)."""
    rng = random.Random(seed)
    for k in range(n_cases):
        variant = rng.choice("UIE")
        beam = rng.choice([1, 2, 3, 5, 8, 16, 16, 24, 32, 40, 64 if variant == "U" else 40])
        gmin = 16 if beam <= 16 else (32 if beam <= 32 else 64)
        case = dict(variant=variant, beam=beam, lm_order=rng.choice([0, 1, 2, 3, 5, 5, 7]), beta=rng.choice([0.0, 0.05, 0.3, 1.0]),
                    margin=rng.choice([0.0, 2.0, 6.0, 20.0, 60.0]), gamma=rng.choice([0.0, 1.0, 3.5]),
                    kappa=rng.choice([0.0, 0.35]), alpha=rng.choice([0.0, 0.6, 1.0]), min_len_for_eos=rng.choice([0, 3, 8]),
                    max_new_tokens=rng.choice([1, 7, 24, 40]), n_words=rng.choice([3, 40, 200, 3000]), lex_seed=rng.randrange(100),
                    max_len=rng.choice([3, 8]), n_corpus=rng.choice([0, 50, 2000]), corpus_seed=rng.randrange(100),
                    n_utts=rng.choice([1, 7, 64, 200]), target_seed=rng.randrange(100), base_seed=rng.randrange(1 << 30),
                    min_words=1, max_words=rng.choice([1, 4]), lanes=rng.choice([0, gmin, 64]), name=f"fuzz{k}",
                    allow_eos_anytime=rng.random() < 0.8)
        if rng.random() < 0.3:                          # digits in the lexicon
            case["alphabet"] = synth.BASE_CHARS[:36]
        if rng.random() < 0.3:                          # words the decoder cannot spell, phrases with a space edge
            case["extra_words"] = rng.sample(["don't", "it's", "zq'x", "new york", "a b", "ice cream cone", "x-ray"], 3)
        if rng.random() < 0.15:
            case["max_new_tokens"] = rng.choice([64, 128])
            case["n_utts"] = min(case["n_utts"], 64)
        yield k, case


def hot_cases(n_cases, seed):
    rng = random.Random(seed)
    for k in range(n_cases):
        variant = rng.choice("UIIE")
        beam = rng.choice([8, 16, 16, 32])
        case = dict(variant=variant, beam=beam, lm_order=rng.choice([5, 5, 5, 3]), beta=rng.choice([-0.3, -0.05, 0.0, 0.05, 0.3, 1.0]),
                    margin=rng.choice([0.0, 2.0, 6.0, 20.0, 60.0]), gamma=rng.choice([0.0, 1.0, 3.5]),
                    kappa=rng.choice([0.0, 0.35]), alpha=rng.choice([0.0, 0.6, 1.0]), min_len_for_eos=rng.choice([0, 3, 8]),
                    max_new_tokens=rng.choice([7, 24, 40, 64]), n_words=rng.choice([3, 40, 200, 3000, 20000]), lex_seed=rng.randrange(100),
                    max_len=rng.choice([3, 8, 12]), n_corpus=rng.choice([0, 50, 2000]), corpus_seed=rng.randrange(100),
                    n_utts=rng.choice([1, 7, 64, 200]), target_seed=rng.randrange(100), base_seed=rng.randrange(1 << 30),
                    min_words=1, max_words=rng.choice([1, 4]), lanes=rng.choice([0, 32, 64]), name=f"hot{k}",
                    allow_eos_anytime=rng.random() < 0.8)
        if rng.random() < 0.4:
            case["alphabet"] = synth.BASE_CHARS[:36]
        if rng.random() < 0.3:
            case["extra_words"] = rng.sample(["don't", "it's", "zq'x", "new york", "a b", "ice cream cone", "x-ray"], 3)
        if case["n_words"] == 20000:                    # (three letters do not make that many distinct words)
            case["n_utts"] = min(case["n_utts"], 64)
            case["max_len"] = max(case["max_len"], 8)
        yield k, case


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    hot = len(sys.argv) > 4 and sys.argv[4] == "hot"
    dev = torch.device("cuda", 0)
    t0 = time.time()
    for k, case in (hot_cases if hot else random_cases)(n_cases, seed):
        if only >= 0 and k != only:
            continue
        ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
        if err is not None:
            assert (ref["lens"] == -1).any(), (case, err)
        else:
            _compare_with_trace(ws, ids, lens, scores, ref, case["name"])
        print(f"{k:3d} ok  {time.time() - t0:6.1f} s  " + " ".join(f"{a}={case[a]}" for a in
              ("variant", "beam", "lm_order", "beta", "margin", "gamma", "max_new_tokens", "n_words", "n_corpus", "n_utts", "lanes")), flush=True)
    print("all", n_cases, "cases equal")
