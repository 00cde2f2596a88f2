"""Diagnostic: run one case of the randomised sweep (fuzz_parity.py numbering) and print the first step whose trace
differs from the C checker, with the lexicon masks and the per-step fp64 scores of that utterance.
usage: debug_case.py <seed> <case> [lanes_per_utterance] [utterance]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np, torch
from oracle import synth
from tests.test_gpu_parity import _run_case_batched

from scripts.fuzz_parity import random_cases

seed, only = int(sys.argv[1]), int(sys.argv[2])
case = [c for k, c in random_cases(only + 1, seed)][-1]
if len(sys.argv) > 3:
    case["lanes"] = int(sys.argv[3])
print(case)
ws, ids, lens, scores, ref, err = _run_case_batched(case, torch.device("cuda", 0))
N, B, T = ws.N, ws.B, ws.T
bp = ws.backptr.cpu().numpy().reshape(max(T, 1), N, B)
tk = ws.token.cpu().numpy().reshape(max(T, 1), N, B)
utt = ws.utt.cpu().numpy()
bad = 0
for u in range(N):
    steps = int(utt[u]) >> 16
    for t in range(min(steps, ref["steps"][u])):
        nb = ref["nbeams"][u, t]
        gt = tk[t, u, :nb].astype(np.int16); gt[gt == 0xFF] = -1
        if not ((bp[t, u, :nb] == ref["parent"][u, t, :nb]).all() and (gt == ref["token"][u, t, :nb]).all()):
            print("utt", u, "step", t, "nb", nb)
            print(" gpu parent", bp[t, u, :nb].tolist()); print(" ref parent", ref["parent"][u, t, :nb].tolist())
            print(" gpu token ", gt.tolist()); print(" ref token ", ref["token"][u, t, :nb].tolist())
            if "score" in ref: print(" ref score", ref["score"][u, t, :nb].tolist())
            if ws.masks is not None:
                mk = ws.masks.cpu().numpy().reshape(max(T, 1), N, B).view(np.uint64)
                nprev = 1 if t == 0 else ref["nbeams"][u, t - 1]
                print(" masks of the parents", [hex(int(x)) for x in mk[t, u, :nprev]])
                print(" ref masks           ", [hex(int(x)) for x in ref["mask"][u, t, :nprev]])
            bad += 1
            break
    if bad >= 3:
        break
print("differing utterances shown:", bad, "ref keys:", list(ref.keys()))

# per-step scores of one utterance: drive the loop by hand
if bad:
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from tests.test_gpu_parity import case_words, case_corpus, case_targets, _pack_targets
    from oracle import ns_oracle as O
    dev = torch.device("cuda", 0)
    words = case_words(case); corpus = case_corpus(case, words)
    lm = O.NgramLM(case["lm_order"], floor_vsz=case["variant"] == "U").fit(corpus) if case["lm_order"] > 0 else None
    tg, tl = _pack_targets(case_targets(case, words))
    dec = NSDecoder(DeviceTables.from_words(words, lm), case["beam"], case["max_new_tokens"], variant=case["variant"],
                    lm_weight=case["beta"], oov_penalty=case["gamma"], word_bonus=case["kappa"],
                    length_penalty_alpha=case["alpha"], min_len_for_eos=case["min_len_for_eos"],
                    lanes_per_utt=case.get("lanes", 0), allow_eos_anytime=case.get("allow_eos_anytime", True))
    w2 = dec.workspace(len(tl), with_masks=True)
    tgd, tld = torch.from_numpy(tg).to(dev), torch.from_numpy(tl).to(dev)
    dec.init(w2)
    uu = int(sys.argv[4]) if len(sys.argv) > 4 else 198
    for t in range(case["max_new_tokens"]):
        dec.synth_logits(w2, case["base_seed"], 0, tgd, tld, case["margin"])
        dec.step(w2)
        torch.cuda.synchronize()
        sc = w2.score.cpu().numpy().reshape(N, B)[uu]
        nb = ref["nbeams"][uu, t] if t < ref["steps"][uu] else 0
        d = [i for i in range(nb) if float(sc[i]).hex() != float(ref["score"][uu, t, i]).hex()]
        if d:
            print("step", t, "first differing rank", d[0], "gpu", sc[:nb].tolist())
            break
