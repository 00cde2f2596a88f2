"""Bindings with the exact signatures of the three copies of ``constrained_beam_decode`` in the reference, one
module per reference file, so that a call site keeps its arguments and only its import line changes:

    scripts/ns_utils.py:194-204      ->  silent_speech_amd.compat.ns_utils
    inference_emg_llm.py:222-231     ->  silent_speech_amd.compat.inference_emg_llm
    eval_ar.py:214-227               ->  silent_speech_amd.compat.eval_ar

Each module also offers ``ids_to_text`` and ``postprocess`` under the reference's names.
"""
