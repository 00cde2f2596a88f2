"""Parity stress for the candidate-buffer overflow path: a lexicon in which every node has all 36 children
(all 1-3 character strings over a-z0-9 up to a count), wide beams and flat logits, so that every pass overflows
the buffer, prunes, and the threshold search runs on more chunks than a lane keeps in registers."""
import itertools, random, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
from oracle import synth
from tests.test_gpu_parity import _compare_with_trace, _run_case_batched

rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
dev = torch.device("cuda", 0)
A = synth.BASE_CHARS[:36]
dense = ["".join(p) for n in (1, 2) for p in itertools.product(A, repeat=n)] + ["".join(p) for p in itertools.product(A[:8], A, A)]
k = 0
for variant in "UIE":
    for beam in (8, 16, 32, 40, 64):
        if variant != "U" and beam > 40:
            continue
        for lanes in sorted({0, 64, 16 if beam <= 16 else (32 if beam <= 32 else 64)}):
            case = dict(variant=variant, beam=beam, lm_order=rng.choice([0, 2, 3, 5, 5]), beta=rng.choice([0.1, 0.1, 0.3, -0.3]),
                        margin=rng.choice([0.0, 1.0]),
                        gamma=1.0, kappa=0.35, alpha=0.6, min_len_for_eos=2, max_new_tokens=10, n_words=3, lex_seed=1,
                        max_len=3, n_corpus=200, corpus_seed=3, n_utts=96, target_seed=5, base_seed=rng.randrange(1 << 30),
                        min_words=1, max_words=3, alphabet=A, extra_words=dense, lanes=lanes, name=f"dense{k}")
            ws, ids, lens, scores, ref, err = _run_case_batched(case, dev)
            assert err is None
            _compare_with_trace(ws, ids, lens, scores, ref, case["name"])
            print(k, "ok", variant, "beam", beam, "lanes", lanes, "order", case["lm_order"], flush=True)
            k += 1
print("all", k, "dense cases equal")
