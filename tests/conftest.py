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
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _error_budget_context(request):
    """names the running test for tests/errbudget.py (measured-error bounds per (test, tensor))"""
    from tests import errbudget
    errbudget.current = request.node.nodeid.split("tests/")[-1]
    yield


def pytest_sessionfinish(session, exitstatus):
    from tests import errbudget
    errbudget.dump()
