import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# The shared-window 3x3 kernel (conv_win.hip) and the fused MLP take a layer only from a size that fills the chip; the parity
# tests run at sizes the oracle finishes in seconds, so those two thresholds are lowered here and every eligible test shape
# exercises the kernels the PRODUCT runs (tests/test_default_dispatch_gpu.py runs mid-size units WITHOUT these overrides, so
# the default selection rules themselves are compared with the oracle).  The library reads them once per process; set them in
# the shell to override.  (The measured-negative kernels of rounds 3-5 — ring convolution, recompute-plan Mlp gradients, folded
# BatchNorm launches, ring-less pointwise GEMM — left the tree in round 6; profiles/ and DESIGN.md keep their numbers.)
os.environ.setdefault('TOK_CONV_WIN_MIN_TILES', '1')
os.environ.setdefault('TOK_CONV_S2D_MIN_TILES', '1')
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
