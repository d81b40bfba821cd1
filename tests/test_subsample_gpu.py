"""Stride-2 projection shortcuts as pointwise layers (GPU): tok_subsample2_fwd / _bwd are exact copies / scatters, and
tok_conv_dgrad_subacc (the pointwise data gradient that absorbs the half-resolution gradient in its accumulate stage) equals
the two-launch form — scatter, then an accumulating tok_conv_dgrad* — bit for bit, in its three epilogue variants, and the
fp32 restatement (tests/fake_backend.py) within bf16 rounding."""
import ctypes

import pytest
import torch

from torchok_amd import _C
from fake_backend import FakeTok
from test_kernels_gpu import BF16, DEV, _desc, relerr, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def libs():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _C.load_library(), FakeTok()


@pytest.mark.parametrize('shape', [(2, 8, 8, 64), (3, 7, 9, 24), (1, 1, 1, 8), (2, 13, 4, 256)])
def test_subsample2_exact(libs, shape):
    lib, _ = libs
    n, h, w, c = shape
    p, q = (h + 1) // 2, (w + 1) // 2
    st = torch.cuda.current_stream().cuda_stream
    x = rnd(n, h, w, c).to(BF16).to(DEV)
    out = torch.empty(n, p, q, c, dtype=BF16, device=DEV)
    assert lib.tok_subsample2_fwd(x.data_ptr(), n, h, w, c, out.data_ptr(), st) == 0, lib.tok_last_error()
    assert torch.equal(out, x[:, ::2, ::2])
    dsub = rnd(n, p, q, c, seed=3).to(BF16).to(DEV)
    dx = torch.full((n, h, w, c), 7.0, dtype=BF16, device=DEV)
    assert lib.tok_subsample2_bwd(dsub.data_ptr(), n, h, w, c, dx.data_ptr(), 0, st) == 0, lib.tok_last_error()
    want = torch.zeros_like(dx)
    want[:, ::2, ::2] = dsub
    assert torch.equal(dx, want)
    old = rnd(n, h, w, c, seed=4).to(BF16).to(DEV)
    dx = old.clone()
    assert lib.tok_subsample2_bwd(dsub.data_ptr(), n, h, w, c, dx.data_ptr(), 1, st) == 0, lib.tok_last_error()
    want = old.clone()
    want[:, ::2, ::2] = (old[:, ::2, ::2].float() + dsub.float()).to(BF16)
    assert torch.equal(dx, want)


# rows >= 100000: the layers the ring kernel serves (TOK_PW_RING_MIN_ROWS); odd extents exercise the ceil(h/2) geometry
SUBACC_CASES = [(8, 112, 112, 64, 64), (8, 113, 111, 128, 64), (2, 225, 223, 64, 256)]


@pytest.mark.parametrize('case', SUBACC_CASES)
@pytest.mark.parametrize('mode', ['plain', 'bnstats', 'bnstats_mask', 'maskstore'])
def test_dgrad_subacc_equals_scatter_then_accumulate(libs, case, mode):
    lib, fake = libs
    n, h, w, c, k = case
    d = _desc(n, h, w, c, k, 1, 1, 0)
    assert lib.tok_conv_dgrad_subacc_ok(ctypes.byref(d)) == 1
    p, q = (h + 1) // 2, (w + 1) // 2
    m = n * h * w
    st = torch.cuda.current_stream().cuda_stream
    dy_h = rnd(n, h, w, k).to(BF16)
    wd_h = rnd(c, 1, 1, k, scale=k ** -0.5).to(BF16)
    dsub_h = rnd(n, p, q, c, seed=2).to(BF16)
    bn_y_h = rnd(n, h, w, c, seed=6).to(BF16)
    mask_h = torch.randint(0, 256, (m, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    dy, wd, dsub, bn_y, mask = (t.to(DEV) for t in (dy_h, wd_h, dsub_h, bn_y_h, mask_h))
    rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
    use_mask = mode in ('bnstats_mask', 'maskstore')
    stats = mode != 'plain'

    # one launch
    dx1 = torch.full((n, h, w, c), float('nan'), dtype=BF16, device=DEV)
    part1 = torch.zeros(2, rows, c, device=DEV)
    rc = lib.tok_conv_dgrad_subacc(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx1.data_ptr(), dsub.data_ptr(),
                                   bn_y.data_ptr() if mode.startswith('bnstats') else None,
                                   mask.data_ptr() if use_mask else None, part1.data_ptr() if stats else None,
                                   1 if mode == 'maskstore' else 0, st)
    assert rc == 0, lib.tok_last_error()
    # two launches
    dx2 = torch.full((n, h, w, c), float('nan'), dtype=BF16, device=DEV)
    part2 = torch.zeros(2, rows, c, device=DEV)
    assert lib.tok_subsample2_bwd(dsub.data_ptr(), n, h, w, c, dx2.data_ptr(), 0, st) == 0
    if mode == 'plain':
        rc = lib.tok_conv_dgrad(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), 1, st)
    elif mode == 'maskstore':
        rc = lib.tok_conv_dgrad_maskstore(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), 1, mask.data_ptr(),
                                          part2.data_ptr(), st)
    else:
        rc = lib.tok_conv_dgrad_bnstats(ctypes.byref(d), dy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), 1, bn_y.data_ptr(),
                                        mask.data_ptr() if use_mask else None, part2.data_ptr(), st)
    assert rc == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert not torch.isnan(dx1.float()).any()
    assert torch.equal(dx1, dx2)
    if stats:
        assert torch.equal(part1, part2)

    # fp32 restatement
    dx_h = torch.zeros(n, h, w, c, dtype=BF16)
    part_h = torch.zeros(2, 2, c)
    rc = fake.tok_conv_dgrad_subacc(d, dy_h.data_ptr(), wd_h.data_ptr(), dx_h.data_ptr(), dsub_h.data_ptr(),
                                    bn_y_h.data_ptr() if mode.startswith('bnstats') else None,
                                    mask_h.data_ptr() if use_mask else None, part_h.data_ptr() if stats else None,
                                    1 if mode == 'maskstore' else 0, None)
    assert rc == 0
    assert relerr(dx1.float(), dx_h.float()) < 5e-3
    if stats:
        got, want = part1.sum(1).cpu(), part_h.sum(1)
        assert relerr(got[0], want[0]) < 4e-3
        if mode != 'maskstore':
            assert relerr(got[1], want[1]) < 4e-3


def test_dgrad_subacc_refuses_layers_off_the_ring(libs):
    lib, _ = libs
    d = _desc(2, 16, 16, 64, 64, 1, 1, 0)          # 512 rows: the two-buffer kernel's territory
    assert lib.tok_conv_dgrad_subacc_ok(ctypes.byref(d)) == 0
    t = torch.zeros(2, 16, 16, 64, dtype=BF16, device=DEV)
    wd = torch.zeros(64, 64, dtype=BF16, device=DEV)
    sub = torch.zeros(2, 8, 8, 64, dtype=BF16, device=DEV)
    rc = lib.tok_conv_dgrad_subacc(ctypes.byref(d), t.data_ptr(), wd.data_ptr(), t.data_ptr(), sub.data_ptr(), None, None, None,
                                   0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b'subacc' in lib.tok_last_error()


@pytest.mark.parametrize('case', [(8, 112, 112, 64, 256, 64), (3, 200, 180, 64, 256, 64), (8, 113, 111, 128, 128, 64)])
@pytest.mark.parametrize('mode', ['plain', 'acc', 'bnstats_mask'])
def test_dgrad2_equals_two_launches(libs, case, mode):
    """dx = dgrad(dy1, w1) + dgrad(dy2, w2) + bias in one ring launch (the fused residual unit's d(input)): the fp32
    restatement within bf16 rounding, the two-launch form (which rounds dx twice) within 1e-2, statistics within 4e-3."""
    lib, fake = libs
    n, h, w, c, k1, k2 = case
    d1, d2 = _desc(n, h, w, c, k1, 1, 1, 0), _desc(n, h, w, c, k2, 1, 1, 0)
    assert lib.tok_conv_dgrad2_ok(ctypes.byref(d1), ctypes.byref(d2)) == 1
    m = n * h * w
    st = torch.cuda.current_stream().cuda_stream
    dy1_h, dy2_h = rnd(m, k1).to(BF16), rnd(m, k2, seed=1).to(BF16)
    w1_h, w2_h = rnd(c, k1, scale=k1 ** -0.5, seed=2).to(BF16), rnd(c, k2, scale=k2 ** -0.5, seed=3).to(BF16)
    bias_h = rnd(c, seed=4)
    old_h = rnd(m, c, seed=5).to(BF16)
    bn_y_h = rnd(m, c, seed=6).to(BF16)
    mask_h = torch.randint(0, 256, (m, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    dy1, dy2, w1, w2, bias, bn_y, mask = (t.to(DEV) for t in (dy1_h, dy2_h, w1_h, w2_h, bias_h, bn_y_h, mask_h))
    acc = 1 if mode == 'acc' else 0
    stats = mode == 'bnstats_mask'
    rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d2))
    dx1 = old_h.to(DEV).clone()
    part1 = torch.zeros(2, rows, c, device=DEV)
    rc = lib.tok_conv_dgrad2(ctypes.byref(d1), dy1.data_ptr(), w1.data_ptr(), ctypes.byref(d2), dy2.data_ptr(), w2.data_ptr(),
                             bias.data_ptr(), dx1.data_ptr(), acc, bn_y.data_ptr() if stats else None,
                             mask.data_ptr() if stats else None, part1.data_ptr() if stats else None, st)
    assert rc == 0, lib.tok_last_error()
    dx2 = old_h.to(DEV).clone()
    part2 = torch.zeros(2, rows, c, device=DEV)
    assert lib.tok_conv_dgrad(ctypes.byref(d1), dy1.data_ptr(), w1.data_ptr(), dx2.data_ptr(), acc, st) == 0
    assert lib.tok_conv_dgrad_bias(ctypes.byref(d2), dy2.data_ptr(), w2.data_ptr(), bias.data_ptr(), dx2.data_ptr(), 1,
                                   bn_y.data_ptr() if stats else None, mask.data_ptr() if stats else None,
                                   part2.data_ptr() if stats else None, st) == 0
    torch.cuda.synchronize()
    ref = dy1_h.float() @ w1_h.float().t() + dy2_h.float() @ w2_h.float().t() + bias_h + (old_h.float() if acc else 0)
    assert relerr(dx1.float(), ref) < 5e-3
    assert relerr(dx1.float(), dx2.float()) < 1e-2
    if stats:
        g1, g2 = part1.sum(1).cpu(), part2.sum(1).cpu()
        bits = ((mask_h.long().unsqueeze(-1) >> torch.arange(8)) & 1).reshape(m, c).float()
        dz = dx1.float().cpu() * bits
        assert relerr(g1[0], dz.sum(0)) < 4e-3 and relerr(g1[1], (dz * bn_y_h.float()).sum(0)) < 4e-3
        assert relerr(g1, g2) < 2e-2
