"""The per-file bindings carry the reference's signatures (names, order, defaults, keyword-only block)."""
import inspect

from silent_speech_amd.compat import eval_ar, inference_emg_llm, ns_utils

P = inspect.Parameter


def _sig(f):
    return [(p.name, p.kind, p.default) for p in inspect.signature(f).parameters.values()]


COMMON = [("step_fn", P.POSITIONAL_OR_KEYWORD, P.empty), ("TT", P.POSITIONAL_OR_KEYWORD, P.empty),
          ("beam", P.POSITIONAL_OR_KEYWORD, P.empty), ("max_new_tokens", P.POSITIONAL_OR_KEYWORD, P.empty),
          ("trie", P.POSITIONAL_OR_KEYWORD, P.empty), ("lm", P.POSITIONAL_OR_KEYWORD, None),
          ("lm_weight", P.POSITIONAL_OR_KEYWORD, 0.3), ("oov_penalty", P.POSITIONAL_OR_KEYWORD, 1.0)]


def test_signatures_equal_the_reference_files():
    # scripts/ns_utils.py:194-204
    assert _sig(ns_utils.constrained_beam_decode) == COMMON + [("allow_eos_anytime", P.POSITIONAL_OR_KEYWORD, True)]
    # inference_emg_llm.py:222-231
    assert _sig(inference_emg_llm.constrained_beam_decode) == COMMON
    # eval_ar.py:214-227
    assert _sig(eval_ar.constrained_beam_decode) == COMMON + [("word_bonus", P.KEYWORD_ONLY, 0.0),
                                                             ("length_penalty_alpha", P.KEYWORD_ONLY, 0.6),
                                                             ("min_len_for_eos", P.KEYWORD_ONLY, 8)]
    for m in (ns_utils, inference_emg_llm, eval_ar):
        assert callable(m.ids_to_text) and callable(m.postprocess)


def test_table_cache_digest_sees_counts_and_inner_trie_nodes():
    """The digest that keys the drop-in's table cache changes with a refit that keeps every dictionary size and
    with a word inserted below an existing first letter; an object without counts gets a constant."""
    from oracle import ns_oracle as O
    from silent_speech_amd.decoder import _fingerprint
    trie = O.build_trie(["ab", "ba"])
    lm = O.NgramLM(2).fit(["ab ba ab ab ab ab ab"])
    f0 = _fingerprint(trie, lm)
    sizes = (len(lm.counts), len(lm.context), len(lm.V))
    lm.fit(["ab ba ba ba ba ba ab"])
    assert (len(lm.counts), len(lm.context), len(lm.V)) == sizes
    f1 = _fingerprint(trie, lm)
    assert f1 != f0 and f1[0] == f0[0]
    trie.insert("aa")
    f2 = _fingerprint(trie, lm)
    assert f2[0] != f1[0] and f2[1] == f1[1]
    assert _fingerprint(trie, None)[1] is None

    class OnlyLogp:
        def logp(self, s):
            return 0.0
    assert _fingerprint(trie, OnlyLogp())[1] == ("?",)
