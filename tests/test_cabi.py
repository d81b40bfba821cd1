"""The C-ABI shared library loads (no GPU needed) and exports every symbol include/tok.h declares."""
import ctypes
import os
import re

from torchok_amd import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'tok.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(tok_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_C.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/tok.h but not exported'
    assert set(names) == set(_C.PROTOTYPES), set(names) ^ set(_C.PROTOTYPES)
    loaded = _C.load_library()
    assert loaded.tok_version() == 1


def test_bad_descriptor_is_reported_not_launched():
    lib = _C.load_library()
    d = _C.ConvDesc(1, 8, 8, 7, 8, 3, 3, 8, 8, 1, 1, 3)     # c = 7: not a multiple of 8
    assert lib.tok_conv_fwd(ctypes.byref(d), 8, 8, None, 8, None, None) == -1
    assert b'multiple of 8' in lib.tok_last_error()
    assert lib.tok_conv_fwd_stat_rows(ctypes.byref(d)) == -1
    d2 = _C.ConvDesc(2, 8, 8, 8, 8, 3, 3, 8, 8, 1, 1, 3)
    assert lib.tok_conv_fwd_stat_rows(ctypes.byref(d2)) >= 1
    assert lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d2)) > 0
    assert lib.tok_bn_bwd_rows(1000, 64) > 0 and lib.tok_bn_stats_rows(10, 7) == -1


def test_missing_library_fails_loudly(tmp_path):
    import pytest
    with pytest.raises(_C.TokError, match='no CPU/eager fallback'):
        _C.load_library(str(tmp_path / 'nope.so'))
