import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


import pytest


@pytest.fixture(autouse=True)
def _seed_torch():
    """Every test starts from the same torch RNG state (a few tests draw stratified jitter with torch.rand)."""
    import torch
    torch.manual_seed(1234)
    yield
