#!/usr/bin/env python3
"""Generate tests/golden/ns_golden_tail.json by running the REFERENCE.

Runs only in the build container, where /root/reference is mounted:

    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/make_golden_tail.py

What is recorded (data only, no reference source text):

  * "scored": the three beam-search copies on the first 64 utterances of the benchmark workload (closed
    200-word lexicon, 5-gram fit on 20 000 sentences, beta 0.05, margin 60, T 64) at beam 8 and beam 16.  Python's
    `sorted` is shadowed in the namespace each copy runs in (the reference files are not touched), so that for
    every pruning step the fp64 scores of the sorted candidates are seen: per utterance the returned ids, the
    transcript, the fp64 score of the returned hypothesis, the smallest gap between the last kept and the first
    dropped candidate over all steps (a decode whose gap is tiny may legitimately differ under another
    log-softmax rounding), the smallest gap between two kept neighbours, a digest of the `step_fn` call
    sequence (= live hypotheses in rank order at every step) and, for the first two utterances, the kept scores
    of every step;
  * "ctc_tail_I" / "ctc_tail_E": the statement blocks of the drivers that turn CTC logits into a proposal and
    choose between it and the AR transcript (inference_emg_llm.py:441-464, eval_ar.py:554-596).  They are inline
    in `main()`; the `if args.use_ctc_rerank:` statement alone is compiled out of the source with `ast` and
    executed with the names it reads (`args`, `model`, `emg`, `TT`, `lm`, `lex_trie`, `lex_words`, `hyp_ar`,
    `step_fn`, ...) supplied here; `main()` is never run;
  * "make_lexicon": the word-frequency / filter / sort / cap / extra-words statements of
    scripts/make_lexicon.py:119-143 (inline in `main()`, extracted the same way) for several inputs;
  * "load_lexicon": scripts/ns_utils.py:68-80 on temporary files.
"""
from __future__ import annotations

import __future__ as _fut
import ast
import base64
import builtins
import hashlib
import json
import os
import sys
import tempfile
import types
import typing
from collections import Counter
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(HERE))

from oracle import synth  # noqa: E402
from oracle.ns_oracle import TextTransformLike  # noqa: E402
from tests.golden_util import corrupt, ctc_frames  # noqa: E402
import make_golden as MG  # noqa: E402


class SortSpy:
    """Stands in for `sorted` where a reference decoder runs: same result, and the keys are seen."""

    def __init__(self):
        self.steps = None

    def __call__(self, it, key=None, reverse=False):
        out = builtins.sorted(it, key=key, reverse=reverse)
        if self.steps is not None and key is not None and out and isinstance(out[0], tuple):
            self.steps.append(([float(key(e)) for e in out], [float(e[0]) for e in out]))
        return out


def scored_cases():
    out = []
    for beam in (8, 16):
        for v in ("U", "I", "E"):
            out.append(dict(name=f"bench64_{v}_b{beam}", variant=v, beam=beam, lm_order=5, max_new_tokens=64,
                            n_words=200, lex_seed=0, min_len=2, max_len=8, n_corpus=20000, corpus_seed=1,
                            n_utts=64, target_seed=1000, min_words=3, max_words=7, base_seed=20260403,
                            margin=60.0, beta=0.05, gamma=1.0, kappa=0.35, alpha=0.6, min_len_for_eos=8))
    # a small margin: no logit dominates its row, so every log-softmax value carries fp32 rounding (with margin
    # 60 the planted token's log-probability is exactly 0 and the others are exact differences)
    for v in ("U", "I", "E"):
        out.append(dict(name=f"soft32_{v}_b16", variant=v, beam=16, lm_order=5, max_new_tokens=48,
                        n_words=200, lex_seed=0, min_len=2, max_len=8, n_corpus=20000, corpus_seed=1,
                        n_utts=32, target_seed=1001, min_words=2, max_words=5, base_seed=77,
                        margin=4.0, beta=0.1, gamma=1.0, kappa=0.35, alpha=0.6, min_len_for_eos=8))
    # the random-access stress shape: 100 000 words over a-z0-9, beam 32 (the first, short cases)
    for v in ("U", "I", "E"):
        out.append(dict(name=f"open4_{v}_b32", variant=v, beam=32, lm_order=5, max_new_tokens=32,
                        n_words=100000, lex_seed=1, alphabet=synth.BASE_CHARS[:36], min_len=2, max_len=12,
                        n_corpus=20000, corpus_seed=1, n_utts=4, target_seed=1002, min_words=2, max_words=4,
                        base_seed=20260403, margin=60.0, beta=0.05, gamma=1.0, kappa=0.35, alpha=0.6,
                        min_len_for_eos=8))
    # the same shape at the benchmark's length: 32 utterances per copy, T 64
    for v in ("U", "I", "E"):
        out.append(dict(name=f"open32_{v}_b32", variant=v, beam=32, lm_order=5, max_new_tokens=64,
                        n_words=100000, lex_seed=1, alphabet=synth.BASE_CHARS[:36], min_len=2, max_len=12,
                        n_corpus=20000, corpus_seed=1, n_utts=32, target_seed=1003, min_words=3, max_words=7,
                        base_seed=20260403, margin=60.0, beta=0.05, gamma=1.0, kappa=0.35, alpha=0.6,
                        min_len_for_eos=8))
    # ... and with a small margin on that lexicon: log-softmax values that carry fp32 rounding where most n-grams
    # of a candidate are unseen
    for v in ("U", "I", "E"):
        out.append(dict(name=f"opensoft16_{v}_b32", variant=v, beam=32, lm_order=5, max_new_tokens=48,
                        n_words=100000, lex_seed=1, alphabet=synth.BASE_CHARS[:36], min_len=2, max_len=12,
                        n_corpus=20000, corpus_seed=1, n_utts=16, target_seed=1004, min_words=2, max_words=5,
                        base_seed=78, margin=4.0, beta=0.1, gamma=1.0, kappa=0.35, alpha=0.6,
                        min_len_for_eos=8))
    # the drivers' own argument defaults (inference_emg_llm.py:313-321: beam 6, 128 steps, 3-gram, weight 0.3, penalty
    # 1.0; eval_ar.py:370-383: beam 6, 64 steps, word bonus 0.35, alpha 0.6, EOS from 8 characters) on a lexicon with
    # phrasebook lines (literal space edges), targets long enough to fill the 128 steps
    for v, T in (("U", 128), ("I", 128), ("E", 64)):
        out.append(dict(name=f"driver_{v}_b6", variant=v, beam=6, lm_order=3, max_new_tokens=T,
                        n_words=200, lex_seed=0, min_len=2, max_len=8,
                        extra_words=["new york", "thank you", "good morning", "ice cream"],
                        n_corpus=20000, corpus_seed=1, n_utts=16, target_seed=1005,
                        min_words=10 if T == 128 else 3, max_words=18 if T == 128 else 7, base_seed=79,
                        margin=12.0, beta=0.3, gamma=1.0, kappa=0.35, alpha=0.6, min_len_for_eos=8))
    return out


def run_scored(ref, spy, case):
    R = ref[case["variant"]]
    words = MG.build_words(case)
    trie = R["build_trie_from_words"](words)
    corpus = synth.make_sentences(words, case["n_corpus"], case["corpus_seed"])
    lm = R["CharNgramLM"](order=case["lm_order"])
    lm.fit(corpus)
    targets = synth.make_sentences(words, case["n_utts"], case["target_seed"], case["min_words"], case["max_words"])
    B = case["beam"]
    utts = []
    for u, tgt in enumerate(targets):
        useed = synth.utt_seed(case["base_seed"], u)
        tok = synth.text_to_tokens(tgt)
        calls = []

        def step_fn(ids, _c=calls, _s=useed, _t=tok):
            _c.append(MG.enc(ids))
            return torch.from_numpy(synth.logits_row(_s, ids, _t, case["margin"])).unsqueeze(0)

        kw = dict(lm=lm, lm_weight=case["beta"], oov_penalty=case["gamma"])
        if case["variant"] == "E":
            kw.update(word_bonus=case["kappa"], length_penalty_alpha=case["alpha"], min_len_for_eos=case["min_len_for_eos"])
        spy.steps = []
        ids = R["constrained_beam_decode"](step_fn, TextTransformLike, B, case["max_new_tokens"], trie, **kw)
        steps, spy.steps = spy.steps, None
        cut = min((k[B - 1] - k[B] for k, _ in steps if len(k) > B), default=float("inf"))
        rank = min((k[i] - k[i + 1] for k, _ in steps for i in range(min(B, len(k)) - 1)), default=float("inf"))
        rec = {"target": tgt, "ids": [int(i) for i in ids], "text": R["ids_to_text"](TextTransformLike, ids),
               "score": float(steps[-1][1][0]).hex(), "steps": len(steps),
               "min_cut_gap": cut if cut != float("inf") else None,
               "min_rank_gap": rank if rank != float("inf") else None,
               "n_calls": len(calls), "calls_sha1": hashlib.sha1("\n".join(calls).encode()).hexdigest()}
        if u < 2:
            kept = [r[:B] for _, r in steps]
            rec["kept_counts"] = [len(k) for k in kept]
            rec["kept_scores_f64"] = base64.b64encode(np.array([x for k in kept for x in k], np.float64).tobytes()).decode()
        assert rec["ids"] == [int(i) for i in ids]
        utts.append(rec)
    return utts


# ----------------------------------------------------------------------
# statement blocks inline in main()
# ----------------------------------------------------------------------
def main_fn(path: Path):
    tree = ast.parse(path.read_text(encoding="utf-8"), filename=str(path))
    return [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]


def rerank_block(path: Path):
    """The `if args.use_ctc_rerank:` statement of the decode loop."""
    hits = [n for n in ast.walk(main_fn(path)) if isinstance(n, ast.If) and isinstance(n.test, ast.Attribute)
            and n.test.attr == "use_ctc_rerank"]
    assert len(hits) == 1, (path, len(hits))
    return hits[0]


def compile_block(nodes, path: Path):
    mod = ast.Module(body=list(nodes), type_ignores=[])
    return compile(mod, str(path), "exec", flags=_fut.annotations.compiler_flag, dont_inherit=True)


class _CtcModel:
    """What the block asks of `model`: adapter(emg) and ctc_head(emb) -> (1, T', V) logits."""

    def adapter(self, emg):
        return emg

    def ctc_head(self, emb):
        return emb


def run_ctc_tail(ref, which: str):
    path = REF / ("inference_emg_llm.py" if which == "I" else "eval_ar.py")
    code = compile_block([rerank_block(path)], path)
    R = ref[which]
    words = synth.make_lexicon(120, 21, max_len=7)
    corpus = synth.make_sentences(words, 1500, 22)
    trie = R["build_trie_from_words"](words)
    lm = R["CharNgramLM"](order=3)
    lm.fit(corpus)
    targets = synth.make_sentences(words, 28, 23, 1, 4)
    beta = 0.3
    rows = []
    for u, tgt in enumerate(targets):
        Tp = 24 + 8 * (u % 9)
        if u == 5:
            tgt_ctc = ""                                    # all blanks: empty proposal
        elif u % 4 == 1:
            tgt_ctc = targets[(u + 1) % len(targets)]       # a different sentence
        else:
            tgt_ctc = tgt
        logits = ctc_frames(500 + u, tgt_ctc, Tp, wrong=0.08 if u % 3 else 0.0)
        hyp_ar_raw = corrupt(700 + u, tgt) if u % 5 else tgt
        if u == 9:
            hyp_ar_raw = ""
        hyp_ar = R["postprocess"](hyp_ar_raw, trie, words)
        useed = synth.utt_seed(4242, u)
        tok = synth.text_to_tokens(tgt)

        def step_fn(ids, _s=useed, _t=tok):
            return torch.from_numpy(synth.logits_row(_s, ids, _t, 7.0)).unsqueeze(0)

        ns = {"args": types.SimpleNamespace(use_ctc_rerank=True, ns_lm_weight=beta), "model": _CtcModel(),
              "emg": torch.from_numpy(logits).unsqueeze(0), "TT": TextTransformLike, "lm": lm, "lex_trie": trie,
              "lex_words": words, "ids_to_text": R["ids_to_text"], "postprocess": R["postprocess"],
              "hyp_ar": hyp_ar, "hyp": hyp_ar, "step_fn": step_fn, "torch": torch, "F": F}
        exec(code, ns)
        rows.append({"target": tgt, "ctc_target": tgt_ctc, "Tp": Tp, "ctc_seed": 500 + u,
                     "wrong": 0.08 if u % 3 else 0.0, "hyp_ar_raw": hyp_ar_raw, "hyp_ar": hyp_ar,
                     "ctc_ids": [int(t) for t in ns["dedup"]], "hyp_ctc": ns["hyp_ctc"], "hyp": ns["hyp"],
                     "scores": [float(ns["cand"][0][1]).hex(), float(ns["cand"][1][1]).hex()]})
    return {"n_words": 120, "lex_seed": 21, "max_len": 7, "n_corpus": 1500, "corpus_seed": 22, "order": 3,
            "target_seed": 23, "n_utts": 28, "beta": beta, "step_seed": 4242, "step_margin": 7.0, "rows": rows}


def run_make_lexicon():
    path = REF / "scripts" / "make_lexicon.py"
    body = [n for n in main_fn(path).body if 119 <= n.lineno < 145]
    code = compile_block(body, path)
    words = synth.make_lexicon(40, 31, max_len=5)
    base_texts = synth.make_sentences(words, 200, 32, 1, 6) + ["naïve café the", "don't stop", "x-ray 42 a1", "  "]
    out = []
    with tempfile.TemporaryDirectory() as td:
        extra = Path(td) / "extra.txt"
        extra_lines = [" Dog ", "cat", "x-ray", "", words[0], "new york", "ÉTÉ", "zz9"]
        extra.write_text("\n".join(extra_lines) + "\n", encoding="utf-8")
        for name, texts, min_freq, min_len, max_words, add in (
                ("defaults", base_texts, 1, 1, 200000, ""),
                ("min_freq3", base_texts, 3, 1, 200000, ""),
                ("min_len3_cap10", base_texts, 1, 3, 10, ""),
                ("extras", base_texts, 2, 1, 15, str(extra)),
                ("cap0_means_none", base_texts[:20], 1, 1, 0, ""),
                ("missing_extra_file", base_texts[:20], 1, 2, 5, str(Path(td) / "nope.txt")),
                ("empty", [], 1, 1, 10, "")):
            ns = {"texts": list(texts), "Counter": Counter, "Path": Path, "TT": TextTransformLike,
                  "args": types.SimpleNamespace(min_freq=min_freq, min_len=min_len, max_words=max_words, add_words=add)}
            exec(code, ns)
            out.append({"name": name, "texts": list(texts), "min_freq": min_freq, "min_len": min_len,
                        "max_words": max_words,
                        "add_words": (extra_lines if add == str(extra) else ([] if add else None)),
                        "lex_words": list(ns["lex_words"])})
    return out


def run_load_lexicon():
    sys.path.insert(0, str(REF / "scripts"))
    import ns_utils
    out = []
    with tempfile.TemporaryDirectory() as td:
        lex, pb = Path(td) / "lexicon.txt", Path(td) / "phrasebook.txt"
        lex_lines = ["the", "The ", "cat", "", "  dog", "zebra", "cat", "don't"]
        pb_lines = ["New York", "ice cream ", "the", ""]
        lex.write_text("\n".join(lex_lines) + "\n", encoding="utf-8")
        pb.write_text("\n".join(pb_lines) + "\n", encoding="utf-8")
        probes = ["the", "th", "new york", "new", "ice cream", "dog", "don't", "don", "", "zebra "]

        def rec(phrasebook, extra):
            trie, words = ns_utils.load_lexicon(str(lex), extra)
            return {"lexicon": lex_lines, "phrasebook": phrasebook, "words": words,
                    "is_word": [[p, bool(trie.is_word(p))] for p in probes],
                    "next_chars": [[p, "".join(sorted(trie.next_chars(p)))] for p in probes]}

        out.append(rec(pb_lines, str(pb)))
        out.append(rec(None, None))
        out.append(rec("missing", str(Path(td) / "nope.txt")))
    return out


def main():
    spy = SortSpy()
    sys.path.insert(0, str(REF / "scripts"))
    import ns_utils
    ref = MG.load_reference()
    ns_utils.sorted = spy                       # module global: shadows the builtin inside ns_utils only
    for v in ("I", "E"):
        ref[v]["sorted"] = spy
    doc = {"meta": {"torch": torch.__version__, "numpy": np.__version__,
                    "cpu_capability": torch.backends.cpu.get_cpu_capability(),
                    "pythonhashseed": os.environ.get("PYTHONHASHSEED", ""), "token_code": MG.TOKCH},
           "scored": []}
    for case in scored_cases():
        c = dict(case)
        c["utts"] = run_scored(ref, spy, case)
        doc["scored"].append(c)
        gaps = [u["min_cut_gap"] for u in c["utts"] if u["min_cut_gap"] is not None]
        print(case["name"], "exact targets", sum(u["text"] == u["target"] for u in c["utts"]),
              "min cut gap", min(gaps) if gaps else None, flush=True)
    doc["ctc_tail_I"] = run_ctc_tail(ref, "I")
    doc["ctc_tail_E"] = run_ctc_tail(ref, "E")
    for k in ("ctc_tail_I", "ctc_tail_E"):
        print(k, sum(r["hyp"] != r["hyp_ar"] for r in doc[k]["rows"]), "of", len(doc[k]["rows"]), "chose the CTC proposal")
    doc["make_lexicon"] = run_make_lexicon()
    doc["load_lexicon"] = run_load_lexicon()
    out = HERE / "ns_golden_tail.json"
    out.write_text(json.dumps(doc, separators=(",", ":"), ensure_ascii=False), encoding="utf-8")
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
