"""MI355X execution engine of torchok_amd (see core.py for the design)."""
import threading
from contextlib import contextmanager

from .core import Region, TTensor, await_ready, mark_padded, pad8, require_device, stream_ptr  # noqa: F401

_tls = threading.local()


def current_region() -> Region:
    r = getattr(_tls, 'region', None)
    if r is None:
        raise RuntimeError('no active torchok_amd region: call the owning backbone/pooling/head module')
    return r


@contextmanager
def region():
    """Open an execution region (one autograd node) for the calling thread."""
    prev = getattr(_tls, 'region', None)
    r = Region()
    _tls.region = r
    try:
        yield r
    finally:
        _tls.region = prev
