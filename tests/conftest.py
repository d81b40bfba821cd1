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
    from torchok_amd import _C
    from fake_backend import FakeTok
    fake = FakeTok()
    prev = _C._install_backend(fake)
    yield fake
    _C._restore_backend(prev)
