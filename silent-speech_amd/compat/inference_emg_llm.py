"""inference_emg_llm.py: the copy its driver calls at :432-435 (no EOS option, unclamped top-k fall-back)."""
from __future__ import annotations

from typing import List

from ..decoder import constrained_beam_decode as _decode
from ..text import ids_to_text, postprocess  # noqa: F401

__all__ = ["constrained_beam_decode", "ids_to_text", "postprocess"]


def constrained_beam_decode(step_fn, TT, beam: int, max_new_tokens: int, trie, lm=None, lm_weight: float = 0.3,
                            oov_penalty: float = 1.0) -> List[int]:
    """Signature of inference_emg_llm.py:222-231."""
    return _decode(step_fn, TT, beam, max_new_tokens, trie, lm, lm_weight, oov_penalty, variant="I")
