import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """torch ships its own HIP runtime (torch/lib/libamdhip64.so) next to the system one that libelfhip.so and
    the facade modules link; torch only finds the GPU when ITS runtime initialises first, so touch it before
    any test can load the facade (which never imports torch)."""
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
