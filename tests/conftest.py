import json
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.loads((ROOT / "tests" / "golden" / "ns_golden.json").read_text(encoding="utf-8"))


@pytest.fixture(scope="session")
def golden_tail():
    """Recorded from the reference by tests/golden/make_golden_tail.py: scores and near-tie gaps at benchmark
    sizes, the CTC proposal / rerank statements of both drivers, make_lexicon's filters, load_lexicon."""
    return json.loads((ROOT / "tests" / "golden" / "ns_golden_tail.json").read_text(encoding="utf-8"))


def pytest_collection_modifyitems(config, items):
    # GPU tests must not be silently skipped on the GPU box and must not run on CPU-only hosts.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
