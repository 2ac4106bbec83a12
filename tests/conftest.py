import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a ROCm device (or without the built extension) reports the GPU tests as skipped
    instead of failing them; `-m gpu` on the GPU box runs them all (a missing libddx.so there is an error, not a skip)."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device on this host")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    # building the checker is not using it: gcc only, a second or two
    from oracle import oracle as orc

    orc.build()
