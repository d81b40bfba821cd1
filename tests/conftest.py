import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The big-tile kernels (conv_ring.hip: 256 x 128 tiles; conv_win.hip: shared-window 3x3) take a layer only from a tile count
# that fills the chip; the parity tests run at sizes the oracle finishes in seconds, so the thresholds are lowered here and
# every eligible test shape exercises them (the full-size property / real-geometry tests reach them at the default thresholds).
# The library reads these once per process; set them in the shell to override.
os.environ.setdefault('TOK_CONV_WIN_MIN_TILES', '1')
os.environ.setdefault('TOK_CONV_RING', '1')            # off by default in the product (measured neutral); kept tested
os.environ.setdefault('TOK_CONV_RING_MIN_TILES', '1')
os.environ.setdefault('TOK_CONV_RING_MIN_K', '64')
os.environ.setdefault('TOK_MLP_MIN_ROWS', '1')          # the fused MLP serves the small test shapes too


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture
def fake_backend():
    """Host-memory stand-in for libtok_gfx950.so (tests/fake_backend.py) — host-logic tests only."""
    import fake_backend as fb
    token = fb.install()
    yield token[0]
    fb.uninstall(token)
