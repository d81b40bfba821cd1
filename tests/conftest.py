import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def fake_backend():
    """Host-memory stand-in for libtok_gfx950.so (tests/fake_backend.py) — host-logic tests only."""
    import fake_backend as fb
    token = fb.install()
    yield token[0]
    fb.uninstall(token)
