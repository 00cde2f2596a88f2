"""nsd_step has two forms of candidate emission, chosen by the launcher from the trie's fan-out (sparse lexicon: rows
emit everything the lexicon allows; dense lexicon: rows that emit after a bound sweep are filtered by its threshold,
and a 4-lanes-per-row pass is split only when it does not fit).  The choice is a tuning guess, so each form must give
the checker's bits on every kind of input: NSD_STEP_DENSE in the environment forces one form for a whole process."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(cmd, dense, timeout=900):
    env = dict(os.environ, NSD_STEP_DENSE=dense)
    return subprocess.run([sys.executable, *cmd], env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("dense", ["0", "1"])
def test_random_cases_with_one_form_forced(dense):
    """Random variants / beams / orders / lexicons (also 20 000-word ones at beams up to 32, negative LM weights)
    against the C checker: parents, tokens, lexicon masks and fp64 scores of every step."""
    out = _run(["scripts/fuzz_parity.py", "120", "4141"], dense)
    assert out.returncode == 0 and "all 120 cases equal" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    out = _run(["scripts/fuzz_parity.py", "80", "4242", "-1", "hot"], dense)
    assert out.returncode == 0 and "all 80 cases equal" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # a lexicon in which every node has all 36 children, wide beams, flat logits: every pass overflows the buffer
    out = _run(["scripts/fuzz_dense.py", "21"], dense)
    assert out.returncode == 0 and "dense cases equal" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("dense", ["0", "1"])
def test_overflowing_and_open_lexicon_cases_with_one_form_forced(dense):
    """The tests whose steps overflow the candidate buffer (100 000-word lexicon, beams 28-60 of the length-normalised
    copy, two utterances per wave overflowing at different times, reference-recorded open-lexicon fixtures)."""
    out = _run(["-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_reference_tail.py", "-m", "gpu", "-x", "-q",
                "--timeout", "300", "--timeout-method", "thread", "-k",
                "open_lexicon or overflowing or largest_lds or edge_shapes or exact_score_ties or idx12 or idx13 or idx14 "
                "or idx15 or idx16 or idx17 or reference_ids_and_scores"], dense)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
