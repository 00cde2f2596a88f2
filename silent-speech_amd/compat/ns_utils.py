"""scripts/ns_utils.py: the copy that offers EOS at every step and clamps the fall-back's k to the vocabulary."""
from __future__ import annotations

from typing import List

from ..decoder import constrained_beam_decode as _decode
from ..text import ids_to_text, postprocess  # noqa: F401

__all__ = ["constrained_beam_decode", "ids_to_text", "postprocess"]


def constrained_beam_decode(step_fn, TT, beam: int, max_new_tokens: int, trie, lm=None, lm_weight: float = 0.3,
                            oov_penalty: float = 1.0, allow_eos_anytime: bool = True) -> List[int]:
    """Signature of scripts/ns_utils.py:194-204."""
    return _decode(step_fn, TT, beam, max_new_tokens, trie, lm, lm_weight, oov_penalty, allow_eos_anytime,
                   variant="U")
