"""MI355X-native NS (NeuroSymbolic) beam decoder of Silent-Speech.

The directory is named ``silent-speech_amd``; it is importable as
``silent_speech_amd`` through the symlink of that name at the repository root.
"""
from .tables import (BASE_CHARS, BOS_ID, EOS_ID, PAD_ID, SPACE_ID, VOCAB, HostTrie, NgramTables,  # noqa: F401
                     compile_ngram, flatten_trie, load_lexicon_words, load_tables, save_tables)
from .text import build_lexicon, finish_transcripts, postprocess, rerank_by_lm  # noqa: F401


class TextTransform:
    """Vocabulary constants of the reference (scripts/data_utils.py:44-50)."""
    BASE_CHARS = BASE_CHARS
    VOCAB_OFFSET = 0
    BOS_IDX = BOS_ID
    EOS_IDX = EOS_ID
    PAD_IDX = PAD_ID
    VOCAB_SIZE = VOCAB


def __getattr__(name):
    # decoder.py needs torch; keep `import silent_speech_amd` light for table-only users
    if name in ("DeviceTables", "NSDecoder", "Workspace", "constrained_beam_decode", "ids_to_text",
                "ctc_greedy_batch", "lm_logp_batch", "rerank_gather_batch", "log_softmax_rows", "rerank_lm_batch",
                "transcribe_batch", "transcribe_batch_joint"):
        from . import decoder
        return getattr(decoder, name)
    raise AttributeError(name)
