import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is the checker of every test: build it once per session."""
    from oracle import oracle

    oracle.build()
