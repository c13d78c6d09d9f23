import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _bounded_cpu_threads():
    """The CPU oracles run small matrices (tiny fixtures) and a few big ones (mini_v0.1 widths): on a many-core GPU host the
    default of one torch thread per core makes the small ones crawl (118 s vs 12 s for the 2580-column tiny restatement)."""
    import torch

    n = os.cpu_count() or 8
    torch.set_num_threads(max(1, min(16, n)))
    yield
