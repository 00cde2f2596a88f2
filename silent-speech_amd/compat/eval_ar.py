"""eval_ar.py: the copy with the word bonus, the boundary EOS and length-normalised pruning (called at :533-545)."""
from __future__ import annotations

from typing import List

from ..decoder import constrained_beam_decode as _decode
from ..text import ids_to_text, postprocess  # noqa: F401

__all__ = ["constrained_beam_decode", "ids_to_text", "postprocess"]


def constrained_beam_decode(step_fn, TT, beam: int, max_new_tokens: int, trie, lm=None, lm_weight: float = 0.3,
                            oov_penalty: float = 1.0, *, word_bonus: float = 0.0,
                            length_penalty_alpha: float = 0.6, min_len_for_eos: int = 8) -> List[int]:
    """Signature of eval_ar.py:214-227."""
    return _decode(step_fn, TT, beam, max_new_tokens, trie, lm, lm_weight, oov_penalty, variant="E",
                   word_bonus=word_bonus, length_penalty_alpha=length_penalty_alpha, min_len_for_eos=min_len_for_eos)
