#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE decoder.

Runs only in the build container, where /root/reference is mounted:

    PYTHONDONTWRITEBYTECODE=1 PYTHONHASHSEED=0 python tests/golden/make_golden.py

What is recorded (data only — no reference source text is stored):

  * for each case: the generator parameters of the lexicon / LM corpus /
    planted targets (oracle/synth.py regenerates them), plus, per utterance,
    the token histories the reference passed to `step_fn` (= its live beams in
    rank order at every step), the lexicon mask it derived for each of them
    (from its `trie.next_chars` / `trie.is_word` calls), the returned ids and
    the transcript;
  * `CharNgramLM.logp` of probe strings for each LM configuration;
  * `postprocess` outputs for sample strings.

The three beam-search copies are taken from
  U: /root/reference/scripts/ns_utils.py          (imported as a module)
  I: /root/reference/inference_emg_llm.py         (definitions extracted)
  E: /root/reference/eval_ar.py                   (definitions extracted)
For I and E the module-level imports need packages that are not installed
(jiwer, unidecode) and a file the reference does not ship
(models/emg_adapter.py), so only the class/function definitions that make up
the NS path are compiled out of the source with `ast` and executed; `main()`
is never run.
"""
from __future__ import annotations

import __future__ as _fut
import ast
import json
import os
import sys
import typing
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

from oracle import synth  # noqa: E402
from oracle.ns_oracle import TextTransformLike  # noqa: E402

WANTED = {"Trie", "build_trie_from_words", "CharNgramLM", "constrained_beam_decode",
          "ids_to_text", "postprocess"}
TOKCH = synth.BASE_CHARS + "^$#"      # printable code for ids 0..39


def load_defs(path: Path) -> dict:
    tree = ast.parse(path.read_text(encoding="utf-8"), filename=str(path))
    body = [n for n in tree.body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in WANTED]
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"torch": torch, "F": F, "__name__": "ref_" + path.stem}
    for k in ("List", "Tuple", "Iterable", "Optional", "Dict"):
        ns[k] = getattr(typing, k)
    code = compile(mod, str(path), "exec", flags=_fut.annotations.compiler_flag, dont_inherit=True)
    exec(code, ns)
    return ns


def load_reference() -> dict:
    sys.path.insert(0, str(REF / "scripts"))
    import ns_utils  # noqa
    return {
        "U": {k: getattr(ns_utils, k) for k in WANTED},
        "I": load_defs(REF / "inference_emg_llm.py"),
        "E": load_defs(REF / "eval_ar.py"),
    }


class TrieSpy:
    """Duck-typed trie that forwards to the reference Trie and records masks."""

    def __init__(self, inner):
        self.inner = inner
        self.masks = []
        self._pending = None

    def next_chars(self, prefix):
        kids = set(self.inner.next_chars(prefix))
        self._pending = kids
        return kids

    def is_word(self, s):
        r = self.inner.is_word(s)
        if self._pending is not None:
            kids, self._pending = self._pending, None
            m = 0
            for c in kids:
                if c != " " and c in synth.BASE_CHARS:
                    m |= 1 << synth.BASE_CHARS.index(c)
            if r or " " in kids:
                m |= 1 << synth.SPACE_ID
            self.masks.append(m)
        return r


def enc(ids):
    return "".join(TOKCH[i] for i in ids[1:])


def build_words(spec):
    words = synth.make_lexicon(spec["n_words"], spec["lex_seed"], spec.get("alphabet", synth.BASE_CHARS[:26]),
                               spec.get("min_len", 2), spec.get("max_len", 6))
    return sorted(set(words + spec.get("extra_words", [])))


def run_case(ref, case):
    R = ref[case["variant"]]
    words = build_words(case)
    trie = R["build_trie_from_words"](words)
    plain = [w for w in words if all(c in synth.BASE_CHARS for c in w)]
    corpus = synth.make_sentences(plain, case["n_corpus"], case["corpus_seed"]) if plain else []
    lm = None
    if case["lm_order"] > 0:
        lm = R["CharNgramLM"](order=case["lm_order"])
        lm.fit(corpus)
    targets = case.get("targets") or synth.make_sentences(plain, case["n_utts"], case["target_seed"],
                                                          case.get("min_words", 2), case.get("max_words", 4))
    utts = []
    for u, tgt in enumerate(targets):
        useed = synth.utt_seed(case["base_seed"], u)
        tgt_tok = [synth.BASE_CHARS.index(c) for c in tgt if c in synth.BASE_CHARS]
        calls = []

        def step_fn(ids, _calls=calls, _useed=useed, _tgt=tgt_tok):
            _calls.append(enc(ids))
            return torch.from_numpy(synth.logits_row(_useed, ids, _tgt, case["margin"])).unsqueeze(0)

        spy = TrieSpy(trie)
        kw = dict(lm=lm, lm_weight=case["beta"], oov_penalty=case["gamma"])
        if case["variant"] == "E":
            kw.update(word_bonus=case["kappa"], length_penalty_alpha=case["alpha"],
                      min_len_for_eos=case["min_len_for_eos"])
        rec = {"target": tgt}
        try:
            ids = R["constrained_beam_decode"](step_fn, TextTransformLike, case["beam"],
                                               case["max_new_tokens"], spy, **kw)
            rec["ids"] = [int(i) for i in ids]
            rec["text"] = R["ids_to_text"](TextTransformLike, ids)
        except RuntimeError as e:          # torch.topk(k > V) in the I/E fall-back
            rec["raises"] = "RuntimeError"
            rec["message"] = str(e)[:80]
        rec["calls"] = calls
        rec["masks"] = [hex(m) for m in spy.masks]
        assert len(calls) == len(spy.masks), (len(calls), len(spy.masks))
        utts.append(rec)
    return utts, words, corpus


def cases():
    out = []
    cid = 0

    def add(**kw):
        nonlocal cid
        base = dict(n_words=60, lex_seed=11, n_corpus=300, corpus_seed=5, n_utts=3, target_seed=100 + cid,
                    base_seed=1000 + cid, margin=12.0, beta=0.05, gamma=1.0, kappa=0.35, alpha=0.6,
                    min_len_for_eos=8, lm_order=5, beam=8, max_new_tokens=20)
        base.update(kw)
        base["name"] = f"c{cid:02d}_{base['variant']}_b{base['beam']}_o{base['lm_order']}"
        out.append(base)
        cid += 1

    for v in ("U", "I", "E"):
        add(variant=v, beam=2, lm_order=5, margin=20.0)
        add(variant=v, beam=8, lm_order=5, margin=60.0, max_new_tokens=28)
        add(variant=v, beam=8, lm_order=3, beta=0.3, margin=8.0)
        add(variant=v, beam=16, lm_order=5, margin=10.0, gamma=0.5, n_utts=2)
        add(variant=v, beam=32, lm_order=5, margin=6.0, max_new_tokens=14, n_utts=2)
        add(variant=v, beam=8, lm_order=0, margin=6.0)
        add(variant=v, beam=6, lm_order=2, beta=0.2, margin=9.0)
        # lexicon with an apostrophe word (node with only a non-BASE child -> fall-back),
        # an accented word and multi-word phrases (literal space edge -> gamma)
        add(variant=v, beam=8, lm_order=5, margin=14.0, n_words=30,
            extra_words=["don't", "été", "new york", "ab cd", "a b", "ne"],
            targets=["don tx ab", "new york ab cd", "a b ne", "ab cdx"], max_new_tokens=16)
        # larger open lexicon over a-z0-9
        add(variant=v, beam=16, lm_order=5, margin=9.0, n_words=5000, lex_seed=3,
            alphabet=synth.BASE_CHARS[:36], max_len=10, n_corpus=2000, n_utts=2, max_new_tokens=18)
        # lexicon with no usable root child: fall-back from the first step
        add(variant=v, beam=4, lm_order=3, n_words=0, extra_words=["'x"], targets=["abc", "de f"],
            max_new_tokens=8, margin=5.0)
        # kappa / eos settings for E (harmless for U/I)
        add(variant=v, beam=8, lm_order=5, margin=11.0, kappa=1.5, alpha=1.0, min_len_for_eos=3,
            max_new_tokens=24)
    # top-k larger than the vocabulary in the fall-back: U clamps, I/E raise
    for v in ("U", "I", "E"):
        add(variant=v, beam=41, lm_order=0, n_words=0, extra_words=["'x"], targets=["ab"],
            max_new_tokens=3, margin=5.0)
    return out


LM_PROBES = ["", " ", "a", "ab", "a ", "abc", "abcd", "abcde", "ab cd", "ab cd ", "zzzzzz", "the quick",
             "q9 x", "$", "^a", "abc  def", " ab", "ABc"]

POST_SAMPLES = ["hellooo   wooorld", "aaab  cc", "", "   ", "xyzzy plugh", "abc"]


def main():
    ref = load_reference()
    doc = {"meta": {"torch": torch.__version__, "numpy": np.__version__,
                    "cpu_capability": torch.backends.cpu.get_cpu_capability(),
                    "pythonhashseed": os.environ.get("PYTHONHASHSEED", ""),
                    "token_code": TOKCH},
           "cases": [], "lm_probes": [], "postprocess": []}
    for case in cases():
        utts, words, corpus = run_case(ref, case)
        c = dict(case)
        c["utts"] = utts
        c["n_lex_words"] = len(words)
        doc["cases"].append(c)
        print(case["name"], [u.get("text", u.get("raises")) for u in utts], flush=True)

    # LM probes, per variant class and order
    words = synth.make_lexicon(60, 11, max_len=6)
    corpus = synth.make_sentences(words, 300, 5)
    for v in ("U", "I", "E"):
        for order in (1, 2, 3, 5, 7):
            lm = ref[v]["CharNgramLM"](order=order)
            lm.fit(corpus)
            probes = LM_PROBES + corpus[:5] + [c[:7] for c in corpus[5:10]]
            doc["lm_probes"].append({"variant": v, "order": order, "n_words": 60, "lex_seed": 11,
                                     "n_corpus": 300, "corpus_seed": 5, "vsz": len(lm.V),
                                     "probes": [[p, float(lm.logp(p)).hex()] for p in probes]})
        lm = ref[v]["CharNgramLM"](order=3)
        lm.fit([])
        doc["lm_probes"].append({"variant": v, "order": 3, "n_words": 0, "lex_seed": 0, "n_corpus": 0,
                                 "corpus_seed": 0, "vsz": len(lm.V),
                                 "probes": [[p, float(lm.logp(p)).hex()] for p in LM_PROBES]})

    # postprocess samples (ns_utils copy)
    trie = ref["U"]["build_trie_from_words"](words)
    for s in POST_SAMPLES + [w + "x" for w in words[:6]] + [w[:-1] for w in words[6:12]]:
        doc["postprocess"].append([s, ref["U"]["postprocess"](s, trie, words)])

    out = HERE / "ns_golden.json"
    out.write_text(json.dumps(doc, separators=(",", ":")), encoding="utf-8")
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
