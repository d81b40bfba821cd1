"""Kernel-level parity (GPU): every C-ABI entry point of libtok_gfx950.so against a plain PyTorch
fp32 restatement of the same op (tests/fake_backend.py, which mirrors the ABI on host memory) on
identical bf16-rounded inputs.  Tolerances: bf16 outputs ≤ 1e-2 relative (north_star), fp32
reductions ≤ 1e-3, index/mask ops bit-exact."""
import ctypes

import pytest
import torch

from torchok_amd import _C
from fake_backend import FakeTok

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = 'cuda'


@pytest.fixture(scope='module')
def libs():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    return _C.load_library(), FakeTok()


def both(libs, name, args_fn):
    """Run entry point `name` natively on device copies and on the host stand-in.
    args_fn(dev) returns the argument list for tensors placed by dev(t)."""
    lib, fake = libs
    host_args = args_fn(lambda t: t)
    dev_tensors = {}

    def to_dev(t):
        d = t.to(DEV)
        dev_tensors[id(t)] = d
        return d
    dev_args = args_fn(to_dev)

    def ptrs(args):
        return [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
    st = torch.cuda.current_stream().cuda_stream
    rc = getattr(lib, name)(*ptrs(dev_args)[:-1], st)
    assert rc == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    rc = getattr(fake, name)(*ptrs(host_args)[:-1], None)
    assert rc == 0
    return dev_tensors


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def maxrel(a, b, floor):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b).abs() / (b.abs() + floor)).max())


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


CONV_CASES = [
    # n, h, w, c, k, r, stride, pad
    (2, 16, 16, 64, 64, 3, 1, 1),
    (2, 16, 16, 64, 128, 1, 1, 0),
    (2, 16, 16, 128, 64, 1, 1, 0),
    (2, 15, 13, 32, 64, 3, 1, 1),      # ragged spatial, C < BK
    (3, 14, 14, 64, 256, 1, 2, 0),     # downsample 1x1 s2
    (2, 17, 19, 64, 128, 3, 2, 1),     # 3x3 s2 ragged
    (2, 8, 8, 8, 24, 3, 1, 1),         # tiny channels, K not multiple of 64
    (1, 7, 7, 512, 512, 3, 1, 1),      # deep K (4608)
    (4, 1, 1, 2048, 1000, 1, 1, 0),    # linear 2048 -> 1000
    (1, 20, 20, 48, 96, 3, 1, 1),      # HRNet-like widths (taps straddle BK)
    (2, 16, 16, 96, 192, 2, 2, 0),     # DaViT patch embed 2x2 s2 (davit.py:62-64)
    (3, 14, 14, 384, 768, 2, 2, 0),
    (32, 8, 8, 128, 256, 1, 2, 0),     # maps smaller than a staging step (ResNet-18 at 64 px: 4x4 outputs, 16 pixels per image)
    (40, 4, 4, 64, 64, 3, 1, 1),
    (70, 2, 2, 64, 128, 3, 2, 1),      # 1x1 outputs: every staged row is another image
    (2, 30, 26, 96, 48, 3, 1, 1),      # 48-wide tiles of the window weight-gradient kernel (HRNet-W48 branches), ragged rows
    (3, 9, 33, 48, 48, 3, 1, 1),
    # shared-window 3x3 kernel (conv_win.hip; served with TOK_CONV_WIN_MIN_TILES=1 at these sizes, see the module fixture)
    (5, 14, 14, 256, 256, 3, 1, 1),    # 16-column tiles, a tile spans two images (flattened rows)
    (3, 28, 28, 128, 128, 3, 1, 1),    # 32-column tiles, 28 of 32 columns used
    (1, 40, 72, 96, 96, 3, 1, 1),      # 64-column tiles, two x-tiles (the second ragged), one 96-channel tile
    (2, 16, 32, 192, 192, 3, 1, 1),    # two 96-channel tiles
    (1, 16, 32, 384, 384, 3, 1, 1),    # few pixel tiles: four 96-channel tiles instead of three 128-channel ones
    (2, 56, 56, 64, 128, 3, 1, 1),
    (5, 13, 15, 40, 136, 3, 1, 1),     # C = 40: second chunk a quarter full; K = 136: second channel tile ragged
    (2, 56, 56, 64, 64, 3, 1, 1),      # 64-channel form of the window kernel (4 x 1 waves): ResNet stage 1
    (1, 40, 72, 48, 48, 3, 1, 1),      # HRNet-W48's high-resolution branch: 48 of 64 channels, C = 32 + 16
    (8, 128, 128, 48, 48, 3, 1, 1),    # ... at a size the register-resident form serves (conv_winr_kernel: 8 x 32 tiles, >= 512 of them)
    (4, 64, 512, 48, 48, 3, 1, 1),     # ... sixteen x-tiles per row group, images of 8 row groups
]


def _desc(n, h, w, c, k, r, stride, pad, s_pad=None):
    p = (h + 2 * pad - r) // stride + 1
    q = (w + 2 * pad - r) // stride + 1
    return _C.ConvDesc(n, h, w, c, k, r, r, p, q, stride, pad, s_pad or r)


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(libs, case):
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    x = rnd(n, h, w, c).to(BF16)
    wt = rnd(k, r, r, c, scale=(r * r * c) ** -0.5).to(BF16)
    bias = rnd(k)
    y = torch.zeros(n, d.p, d.q, k, dtype=BF16)
    # each side lays its partial sums out as [2][its own row count][k]: the buffer holds the larger count, the device result is
    # read back with the library's row count, and only column sums are compared
    rows_dev = libs[0].tok_conv_fwd_stat_rows(ctypes.byref(d))
    stats = torch.zeros(2 * max(rows_dev, libs[1].tok_conv_fwd_stat_rows(d)) * k)
    dv = both(libs, 'tok_conv_fwd', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                               f(x), f(wt), f(bias), f(y), f(stats), None])
    yd, sd = dv[id(y)], dv[id(stats)][:2 * rows_dev * k].view(2, rows_dev, k)
    assert relerr(yd.float(), y.float()) < 4e-3
    assert maxrel(yd.float(), y.float(), 0.05) < 3e-2
    # the partial sums are taken over the fp32 accumulators (before the bf16 store): against the
    # sums of the stored values they may differ by the rounding noise, <= 2^-8 relative per element
    f = yd.float().reshape(-1, k)
    assert relerr(sd[0].sum(0), f.sum(0)) < 4e-3 or float((sd[0].sum(0).cpu() - f.sum(0).cpu()).abs().max()) < 2e-2
    assert relerr(sd[1].sum(0), (f * f).sum(0)) < 4e-3


def test_conv_fwd_stem_c4(libs):
    n, h, w, k = 2, 33, 35, 64
    d = _desc(n, h, w, 4, k, 7, 2, 3, s_pad=8)
    x = rnd(n, h, w, 4).to(BF16)
    x[..., 3] = 0
    wt = rnd(k, 7, 8, 4, scale=0.08).to(BF16)
    wt[:, :, 7, :] = 0
    wt[..., 3] = 0
    y = torch.zeros(n, d.p, d.q, k, dtype=BF16)
    rows = max(libs[0].tok_conv_fwd_stat_rows(ctypes.byref(d)), libs[1].tok_conv_fwd_stat_rows(d))
    stats = torch.zeros(2 * rows * k)
    dv = both(libs, 'tok_conv_fwd', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                               f(x), f(wt), None, f(y), f(stats), None])
    assert relerr(dv[id(y)].float(), y.float()) < 4e-3


@pytest.mark.parametrize('n,h,w,k', [(4, 70, 72, 64), (3, 64, 96, 32), (2, 224, 224, 64)])
def test_conv_fwd_stem_on_the_shared_window(libs, n, h, w, k):
    """csrc/stem.hip (round 5): the 7x7 / stride 2 / padding 3 stem of a 4-channel-padded image on a shared input window —
    served for even widths from 16 tiles up; ragged tile edges (35 x 36 outputs), fewer than 64 output channels, the real
    224 x 224 geometry.  Output and BatchNorm partial sums against the fp32 restatement; the sums are those of the STORED values."""
    lib, fake = libs
    d = _desc(n, h, w, 4, k, 7, 2, 3, s_pad=8)
    x = rnd(n, h, w, 4).to(BF16)
    x[..., 3] = 0
    wt = rnd(k, 7, 8, 4, scale=0.08).to(BF16)
    wt[:, :, 7, :] = 0
    wt[..., 3] = 0
    y = torch.zeros(n, d.p, d.q, k, dtype=BF16)
    rows = lib.tok_conv_fwd_stat_rows(ctypes.byref(d))
    stats = torch.full((2, rows, k), 7.0)
    stats_h = torch.zeros(2 * fake.tok_conv_fwd_stat_rows(d) * k)
    yd, sd = y.to(DEV), stats.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tok_conv_fwd(ctypes.byref(d), x.to(DEV).data_ptr(), wt.to(DEV).data_ptr(), None, yd.data_ptr(), sd.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert fake.tok_conv_fwd(d, x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), stats_h.data_ptr(), None) == 0
    assert relerr(yd.float(), y.float()) < 4e-3
    f = yd.float().reshape(-1, k)
    assert relerr(sd[0].sum(0), f.sum(0)) < 1e-4 or float((sd[0].sum(0).cpu() - f.sum(0).cpu()).abs().max()) < 2e-2
    assert relerr(sd[1].sum(0), (f * f).sum(0)) < 1e-4
    # bit-reproducible, and without statistics the same output
    y2 = torch.zeros_like(yd)
    assert lib.tok_conv_fwd(ctypes.byref(d), x.to(DEV).data_ptr(), wt.to(DEV).data_ptr(), None, y2.data_ptr(), None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y2, yd)


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[3] % 8 == 0])
@pytest.mark.parametrize('accumulate', [0, 1])
def test_conv_dgrad(libs, case, accumulate):
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    dy = rnd(n, d.p, d.q, k).to(BF16)
    wd = rnd(c, r, r, k, scale=(r * r * k) ** -0.5).to(BF16)
    dx = rnd(n, h, w, c, seed=5).to(BF16)
    dv = both(libs, 'tok_conv_dgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                 f(dy), f(wd), f(dx), accumulate, None])
    assert relerr(dv[id(dx)].float(), dx.float()) < 5e-3


S2D_CASES = [(3, 56, 56, 128, 128, 3, 2, 1),    # 28 x 28 class maps: 32-wide tiles, 128 channels
             (5, 28, 28, 256, 256, 3, 2, 1),    # 14 x 14: 16-wide tiles, rows of several images in one tile
             (1, 64, 256, 48, 48, 3, 2, 1),     # wide map walked in x-tiles, 64-channel tiles, 48 = 32 + a 16-channel tail
             (2, 32, 128, 96, 192, 3, 2, 1),    # one 96-channel tile
             (2, 32, 64, 192, 192, 3, 2, 1),    # two 96-channel tiles
             (3, 36, 44, 64, 40, 3, 2, 1)]      # ragged: 18 x 22 class maps, 40 gathered channels


@pytest.mark.parametrize('case', S2D_CASES)
@pytest.mark.parametrize('mode', ['plain', 'accumulate', 'bnstats', 'bnstats_mask'])
def test_conv_dgrad_stride2_on_the_shared_window(libs, case, mode):
    """conv_s2d.hip: the four parity classes of a stride-2 data gradient on one window of dY, every epilogue it serves,
    against the restatement (tests/fake_backend.py)."""
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    lib, fake = libs
    dy = rnd(n, d.p, d.q, k).to(BF16)
    wd = rnd(c, r, r, k, scale=(r * r * k) ** -0.5).to(BF16)
    dx = rnd(n, h, w, c, seed=5).to(BF16)
    if mode in ('plain', 'accumulate'):
        dv = both(libs, 'tok_conv_dgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                     f(dy), f(wd), f(dx), int(mode == 'accumulate'), None])
        assert relerr(dv[id(dx)].float(), dx.float()) < 5e-3
        return
    with_mask = mode == 'bnstats_mask'
    bn_y = rnd(n, h, w, c, seed=6).to(BF16)
    mask = torch.randint(0, 256, (n * h * w, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
    assert rows > 0
    part_d = torch.full((2, rows, c), 7.0)          # every row must be written
    part_h = torch.zeros(2, 2, c)
    dv = both(libs, 'tok_conv_dgrad_bnstats',
              lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d, f(dy), f(wd), f(dx), 0, f(bn_y),
                         f(mask) if with_mask else None, f(part_d) if f.__name__ == 'to_dev' else f(part_h), None])
    assert relerr(dv[id(dx)].float(), dx.float()) < 5e-3
    g = dv[id(dx)].float().cpu().reshape(-1, c)
    bits = ((mask.long().unsqueeze(-1) >> torch.arange(8)) & 1).reshape(-1, c).float() if with_mask else 1.0
    dz = g * bits
    got = dv[id(part_d)].cpu().sum(1)
    assert relerr(got[0], dz.sum(0)) < 4e-3
    assert relerr(got[1], (dz * bn_y.float().reshape(-1, c)).sum(0)) < 4e-3


@pytest.mark.parametrize('case', S2D_CASES + [(2, 20, 36, 48, 48, 3, 1, 1), (1, 24, 40, 96, 96, 3, 1, 1),
                                              (2, 16, 32, 192, 192, 3, 1, 1), (2, 16, 32, 64, 64, 3, 1, 1),
                                              (8, 128, 128, 48, 48, 3, 1, 1)])
@pytest.mark.parametrize('mode', ['maskstore', 'bias_sums'])
def test_conv_dgrad_window_kernels_mask_store_and_bias(libs, case, mode):
    """The remaining epilogues of the window kernels (conv_win.hip / conv_s2d.hip) on every channel-tile form (48 = half +
    quarter, 64, 96 = three halves, 128): the masked store with its one sum, and bias + accumulate + BatchNorm-backward sums."""
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    lib, fake = libs
    dy = rnd(n, d.p, d.q, k).to(BF16)
    wd = rnd(c, r, r, k, scale=(r * r * k) ** -0.5).to(BF16)
    dx = rnd(n, h, w, c, seed=5).to(BF16)
    mask = torch.randint(0, 256, (n * h * w, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    bits = ((mask.long().unsqueeze(-1) >> torch.arange(8)) & 1).reshape(-1, c).float()
    rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
    part_d = torch.full((2, rows, c), 7.0)
    part_h = torch.zeros(2, 2, c)
    dref = lambda f: ctypes.byref(d) if f.__name__ == 'to_dev' else d        # noqa: E731
    pref = lambda f: f(part_d) if f.__name__ == 'to_dev' else f(part_h)      # noqa: E731
    if mode == 'maskstore':
        dv = both(libs, 'tok_conv_dgrad_maskstore', lambda f: [dref(f), f(dy), f(wd), f(dx), 1, f(mask), pref(f), None])
        got = dv[id(dx)].float().cpu().reshape(-1, c)
        assert relerr(got, dx.float().reshape(-1, c)) < 5e-3
        assert float((got * (1 - bits)).abs().max()) == 0.0                  # masked positions are stored as zeros
        assert relerr(dv[id(part_d)].cpu()[0].sum(0), got.sum(0)) < 4e-3
        return
    bias = rnd(c, seed=11)
    bn_y = rnd(n, h, w, c, seed=6).to(BF16)
    dv = both(libs, 'tok_conv_dgrad_bias', lambda f: [dref(f), f(dy), f(wd), f(bias), f(dx), 1, f(bn_y), f(mask), pref(f), None])
    got = dv[id(dx)].float().cpu().reshape(-1, c)
    assert relerr(got, dx.float().reshape(-1, c)) < 5e-3
    dz = got * bits
    sums = dv[id(part_d)].cpu().sum(1)
    assert relerr(sums[0], dz.sum(0)) < 4e-3
    assert relerr(sums[1], (dz * bn_y.float().reshape(-1, c)).sum(0)) < 4e-3


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_wgrad(libs, case):
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    x = rnd(n, h, w, c).to(BF16)
    dy = rnd(n, d.p, d.q, k).to(BF16)
    dw = torch.zeros(k, r, r, c)
    wsb = libs[0].tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.zeros(max(wsb // 4, 16))
    dv = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                 f(x), f(dy), f(dw), k, c, f(ws), wsb, 0, None])
    assert relerr(dv[id(dw)], dw) < 2e-3
    # determinism: a second run is bit-identical (fixed-order partial reduction, no atomics)
    first = dv[id(dw)].clone()
    dv2 = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                  f(x), f(dy), f(dw), k, c, f(ws), wsb, 0, None])
    assert torch.equal(first, dv2[id(dw)])


# token matrices with a bias ([timm] Mlp fc1/fc2, WindowAttention qkv/proj of swin_v2.py; k_real < k_pad: the 1000-class head)
WGRAD_BIAS_CASES = [(64, 56, 56, 96, 288, 288), (4, 7, 7, 768, 3072, 3072), (3, 1, 1, 2048, 1000, 1000),
                    (2, 28, 28, 192, 192, 192), (1, 5, 3, 64, 64, 64), (2, 14, 14, 384, 1536, 1530)]


@pytest.mark.parametrize('case', WGRAD_BIAS_CASES)
@pytest.mark.parametrize('acc', [0, 1])
def test_conv_wgrad_bias(libs, case, acc):
    """dW and the bias gradient (column sums of dy) from one launch: equal to tok_conv_wgrad bit for bit on dW, and to the
    fp32 column sums of the bf16 dy within fp32 summation noise."""
    lib, _ = libs
    n, h, w, c, kpad, k = case
    kpad = (kpad + 7) // 8 * 8
    d = _desc(n, h, w, c, kpad, 1, 1, 0)
    assert lib.tok_conv_wgrad_bias_ok(ctypes.byref(d)) == 1
    x = rnd(n, h, w, c).to(BF16)
    dy = rnd(n, h, w, kpad).to(BF16)
    dw = rnd(k, 1, 1, c, seed=3)
    db = rnd(k, seed=4)
    wsb = lib.tok_conv_wgrad_bias_ws_bytes(ctypes.byref(d))
    assert wsb >= lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d)) + 4 * k
    ws = torch.zeros(max(wsb // 4, 16))
    dv = both(libs, 'tok_conv_wgrad_bias', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                      f(x), f(dy), f(dw), k, c, f(ws), wsb, acc, f(db), acc, None])
    assert relerr(dv[id(dw)], dw) < 2e-3
    assert relerr(dv[id(db)], db) < 1e-4
    dw2 = rnd(k, 1, 1, c, seed=3)
    dv2 = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                  f(x), f(dy), f(dw2), k, c, f(ws), wsb, acc, None])
    assert torch.equal(dv[id(dw)], dv2[id(dw2)])


def test_conv_wgrad_bias_refuses_layers_off_the_ring(libs):
    lib, _ = libs
    d = _desc(2, 16, 16, 64, 64, 3, 1, 1)
    assert lib.tok_conv_wgrad_bias_ok(ctypes.byref(d)) == 0
    t = torch.zeros(16, device=DEV)
    rc = lib.tok_conv_wgrad_bias(ctypes.byref(d), t.data_ptr(), t.data_ptr(), t.data_ptr(), 64, 64, t.data_ptr(), 64, 0,
                                 t.data_ptr(), 0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b'wgrad_bias' in lib.tok_last_error()


def test_conv_wgrad_stem_c4(libs):
    n, h, w, k = 2, 33, 35, 64
    d = _desc(n, h, w, 4, k, 7, 2, 3, s_pad=8)
    x = rnd(n, h, w, 4).to(BF16)
    x[..., 3] = 0
    dy = rnd(n, d.p, d.q, k).to(BF16)
    dw = torch.zeros(k, 7, 7, 3)
    wsb = libs[0].tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.zeros(max(wsb // 4, 16))
    dv = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                 f(x), f(dy), f(dw), k, 3, f(ws), wsb, 0, None])
    assert relerr(dv[id(dw)], dw) < 2e-3


@pytest.mark.parametrize('n,h,w,k', [(4, 70, 72, 64), (3, 64, 96, 32), (2, 224, 224, 64)])
@pytest.mark.parametrize('acc', [0, 1])
def test_conv_wgrad_stem_on_the_shared_window(libs, n, h, w, k, acc):
    """csrc/stem.hip (round 5): the stem's weight gradient with both MFMA operands read by transpose reads — dy from an LDS
    tile, the patches straight out of the staged input window.  Even widths, ragged tiles, fewer than 64 filters, the real
    geometry; overwrite / accumulate; against the fp32 restatement, and bit-reproducible."""
    lib, fake = libs
    d = _desc(n, h, w, 4, k, 7, 2, 3, s_pad=8)
    x = rnd(n, h, w, 4).to(BF16)
    x[..., 3] = 0
    dy = rnd(n, d.p, d.q, k).to(BF16)
    dw = rnd(k, 7, 7, 3, seed=3)
    wsb = lib.tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.zeros(max(wsb // 4, 16))
    dv = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                 f(x), f(dy), f(dw), k, 3, f(ws), wsb, acc, None])
    assert relerr(dv[id(dw)], dw) < 2e-3
    dw2 = rnd(k, 7, 7, 3, seed=3).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tok_conv_wgrad(ctypes.byref(d), dv[id(x)].data_ptr(), dv[id(dy)].data_ptr(), dw2.data_ptr(), k, 3,
                              dv[id(ws)].data_ptr(), wsb, acc, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dw2, dv[id(dw)])


def test_pack_weights(libs):
    k, r, s, c = 10, 3, 3, 3
    src = rnd(k, r, s, c)
    dst = torch.zeros(16, r, 8, 4, dtype=BF16)
    dv = both(libs, 'tok_pack_weight_fwd', lambda f: [f(src), k, r, s, c, f(dst), 16, 8, 4, None])
    assert torch.equal(dv[id(dst)].cpu(), dst)
    dst2 = torch.zeros(8, r, s, 16, dtype=BF16)
    dv = both(libs, 'tok_pack_weight_dgrad', lambda f: [f(src), k, r, s, c, f(dst2), 16, 8, None])
    assert torch.equal(dv[id(dst2)].cpu(), dst2)


@pytest.mark.parametrize('dt', [torch.float32, torch.float16, torch.bfloat16])
def test_nchw_to_nhwc(libs, dt):
    n, c, h, w = 2, 3, 9, 11
    src = rnd(n, c, h, w).to(dt)
    code = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}[dt]
    for cp in (4, 8):
        dst = torch.ones(n, h, w, cp, dtype=BF16)
        dv = both(libs, 'tok_nchw_to_nhwc_bf16', lambda f: [f(src), code, n, c, h, w, f(dst), cp, None])
        assert torch.equal(dv[id(dst)].cpu(), dst)


@pytest.mark.parametrize('m,c', [(1000, 64), (4096 + 17, 256), (300, 2048), (777, 48)])
@pytest.mark.parametrize('relu,with_sc', [(1, 0), (1, 1), (0, 0)])
def test_bn_chain(libs, m, c, relu, with_sc):
    """finalize -> act_fwd -> bwd_reduce -> bwd_finalize -> bwd_apply against the host restatement."""
    lib, fake = libs
    y = (rnd(m, c) * 1.5 + 0.3).to(BF16)
    f = y.float()
    rows = 3
    stats = torch.zeros(2, rows, c)
    stats[0, 0], stats[1, 0] = f[: m // 2].sum(0), (f[: m // 2] ** 2).sum(0)
    stats[0, 2], stats[1, 2] = f[m // 2:].sum(0), (f[m // 2:] ** 2).sum(0)
    gamma, beta = rnd(c, seed=1) * 0.5 + 1, rnd(c, seed=2) * 0.2
    rm, rv, nbt = rnd(c, seed=3), rnd(c, seed=4).abs() + 0.5, torch.tensor([7])
    mean, rstd, scale, shift = (torch.zeros(c) for _ in range(4))
    dv = both(libs, 'tok_bn_finalize', lambda fn: [fn(stats), rows, m, c, c, fn(gamma), fn(beta), fn(rm), fn(rv),
                                                   fn(nbt), 0.1, 1e-5, fn(mean), fn(rstd), fn(scale), fn(shift), None])
    for t in (mean, rstd, scale, shift, rm, rv):
        assert maxrel(dv[id(t)], t, 1e-3) < 1e-4
    assert int(dv[id(nbt)].item()) == 8 == int(nbt.item())
    # torch's own training BatchNorm as the independent check of the statistics semantics
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(gamma), bn.bias.copy_(beta)
    ref = bn(f.t().reshape(1, c, m, 1)).reshape(c, m).t()
    assert relerr(f * scale + shift, ref) < 1e-4

    sc = rnd(m, c, seed=9).to(BF16) if with_sc else None
    out = torch.zeros(m, c, dtype=BF16)
    mask = torch.zeros(m, c // 8, dtype=torch.uint8)
    dv = both(libs, 'tok_bn_act_fwd', lambda fn: [fn(y), fn(scale), fn(shift), fn(sc) if with_sc else None, relu,
                                                  fn(out), fn(mask), m, c, None])
    assert relerr(dv[id(out)].float(), out.float()) < 1e-3
    assert (dv[id(out)].cpu() != out).float().mean() < 1e-3  # fma vs mul+add may flip a last bit
    # the bit mask is exactly (out > 0) of the kernel's own output
    got_bits = ((dv[id(mask)].cpu().long().unsqueeze(-1) >> torch.arange(8)) & 1).reshape(m, c).bool()
    assert torch.equal(got_bits, dv[id(out)].cpu().float() > 0)

    dout = rnd(m, c, seed=11).to(BF16)
    rows_b = lib.tok_bn_bwd_rows(m, c)
    part_d = torch.zeros(2, rows_b, c, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = dict(dout=dout, y=y, out=mask if (relu and with_sc) else None)   # `out` slot = ReLU bit mask
    dd = {k_: (v.to(DEV) if v is not None else None) for k_, v in args.items()}
    cd = {k_: v.to(DEV) for k_, v in dict(scale=scale, shift=shift, mean=mean, rstd=rstd, gamma=gamma).items()}
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    assert lib.tok_bn_bwd_reduce(P(dd['dout']), P(dd['y']), P(dd['out']), P(cd['scale']), P(cd['shift']),
                                 P(cd['mean']), P(cd['rstd']), relu, m, c, P(part_d), st) == 0
    part_h = torch.zeros(2, 1, c)
    assert fake.tok_bn_bwd_reduce(P(dout), P(y), P(args['out']), P(scale), P(shift), P(mean), P(rstd), relu, m, c,
                                  P(part_h), None) == 0
    torch.cuda.synchronize()
    assert relerr(part_d.sum(1), part_h.sum(1)) < 1e-3

    dg_d, db_d, coef_d = (torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(3, c, device=DEV))
    dg_h, db_h, coef_h = torch.zeros(c), torch.zeros(c), torch.zeros(3, c)
    assert lib.tok_bn_bwd_finalize(P(part_d), rows_b, m, c, c, P(cd['gamma']), P(cd['mean']), P(cd['rstd']), P(dg_d),
                                   P(db_d), P(coef_d), 0, 0, st) == 0
    assert fake.tok_bn_bwd_finalize(P(part_h), 1, m, c, c, P(gamma), P(mean), P(rstd), P(dg_h), P(db_h), P(coef_h), 0,
                                    0, None) == 0
    torch.cuda.synchronize()
    assert relerr(dg_d, dg_h) < 1e-3 and relerr(db_d, db_h) < 1e-3 and relerr(coef_d, coef_h) < 1e-3

    dy_d = torch.zeros(m, c, dtype=BF16, device=DEV)
    ds_d = torch.zeros(m, c, dtype=BF16, device=DEV) if with_sc else None
    dy_h = torch.zeros(m, c, dtype=BF16)
    ds_h = torch.zeros(m, c, dtype=BF16) if with_sc else None
    coef_same = coef_h.to(DEV)
    assert lib.tok_bn_bwd_apply(P(dd['dout']), P(dd['y']), P(dd['out']), P(cd['scale']), P(cd['shift']), P(coef_same),
                                relu, P(dy_d), P(ds_d), 0, m, c, st) == 0
    assert fake.tok_bn_bwd_apply(P(dout), P(y), P(args['out']), P(scale), P(shift), P(coef_h), relu, P(dy_h), P(ds_h),
                                 0, m, c, None) == 0
    torch.cuda.synchronize()
    assert relerr(dy_d.float(), dy_h.float()) < 2e-3
    if with_sc:
        assert torch.equal(ds_d.cpu(), ds_h)  # masked copy of the incoming gradient: exact
    # independent check of the whole BN backward against autograd (fp32)
    yy = f.clone().requires_grad_(True)
    z = torch.nn.functional.batch_norm(yy, None, None, gamma, beta, True, 0.1, 1e-5)
    if with_sc:
        z = z + sc.float()
    if relu:
        z = z.relu()
    z.backward(dout.float())
    assert relerr(dy_h.float(), yy.grad) < 2e-2


@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (1, 15, 17, 8), (3, 7, 9, 128)])
def test_maxpool_exact(libs, shape):
    n, h, w, c = shape
    x = rnd(n, h, w, c).relu().to(BF16)   # post-ReLU input: many exact ties at 0
    p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.zeros(n, p, q, c, dtype=BF16)
    am = torch.zeros(n, p, q, c, dtype=torch.uint8)
    dv = both(libs, 'tok_maxpool3x3s2_fwd', lambda f: [f(x), f(y), f(am), n, h, w, c, None])
    assert torch.equal(dv[id(y)].cpu(), y)
    assert torch.equal(dv[id(am)].cpu(), am)
    dy = rnd(n, p, q, c, seed=3).to(BF16)
    for acc in (0, 1):
        dx = rnd(n, h, w, c, seed=4).to(BF16)
        dv = both(libs, 'tok_maxpool3x3s2_bwd', lambda f: [f(dy), f(am), f(dx), acc, n, h, w, c, None])
        assert relerr(dv[id(dx)].float(), dx.float()) < 2e-3
        if not acc:
            assert torch.equal(dv[id(dx)].cpu() != 0, dx != 0)  # gradient routing is index-exact


def test_gap_colsum(libs):
    n, hw, c = 5, 49, 256
    x = rnd(n, hw, c).to(BF16)
    y = torch.zeros(n, c, dtype=BF16)
    dv = both(libs, 'tok_gap_fwd', lambda f: [f(x), f(y), n, hw, c, None])
    assert relerr(dv[id(y)].float(), y.float()) < 2e-3
    dy = rnd(n, c, seed=1).to(BF16)
    for acc in (0, 1):
        dx = rnd(n, hw, c, seed=2).to(BF16)
        dv = both(libs, 'tok_gap_bwd', lambda f: [f(dy), f(dx), acc, n, hw, c, None])
        assert relerr(dv[id(dx)].float(), dx.float()) < 2e-3
    m, npad, nreal = 300, 16, 10
    g = rnd(m, npad).to(BF16)
    out = torch.ones(nreal)
    dv = both(libs, 'tok_colsum', lambda f: [f(g), m, npad, nreal, f(out), 1, None])
    assert relerr(dv[id(out)], out) < 1e-4


@pytest.mark.parametrize('rows,classes', [(256, 1000), (7, 10), (64, 11318), (70001, 19), (20000, 64), (16384, 3), (16500, 33)])
def test_softmax_ce(libs, rows, classes):
    ld = (classes + 7) // 8 * 8
    z = (rnd(rows, ld) * 3).to(BF16)
    t = torch.randint(0, classes, (rows,), generator=torch.Generator().manual_seed(1))
    t[::5] = -100
    lse, rl, loss = torch.zeros(rows), torch.zeros(rows), torch.zeros(_C.TOK_CE_LOSS_FLOATS)
    dv = both(libs, 'tok_softmax_ce_fwd', lambda f: [f(z), f(t), rows, classes, ld, -100, f(lse), f(rl), f(loss), None])
    assert maxrel(dv[id(lse)], lse, 1e-3) < 1e-4
    assert maxrel(dv[id(loss)][:2], loss[:2], 1e-6) < 1e-5
    ref = torch.nn.functional.cross_entropy(z.float()[:, :classes], t, ignore_index=-100)
    assert abs(float(dv[id(loss)][0]) - float(ref)) < 1e-4 * max(1.0, abs(float(ref)))
    assert float(dv[id(loss)][1]) == float((t != -100).sum())
    gs = torch.tensor([0.7])
    dl = torch.ones(rows, ld, dtype=BF16)
    dv = both(libs, 'tok_softmax_ce_bwd', lambda f: [f(z), f(t), f(lse), f(loss), f(gs), rows, classes, ld, -100,
                                                     f(dl), None])
    assert relerr(dv[id(dl)].float(), dl.float()) < 3e-3
    assert torch.all(dv[id(dl)][:, classes:] == 0) and torch.all(dv[id(dl)][::5] == 0)


def test_optimizers_match_torch(libs):
    """tok_sgd_step / tok_adam_step against torch.optim (the optimizers the reference registers)."""
    lib = libs[0]
    n = 10007
    st = torch.cuda.current_stream().cuda_stream
    p0, grads = rnd(n), [rnd(n, seed=s) for s in range(1, 4)]
    for nesterov, wd, damp in ((False, 1e-4, 0.0), (True, 0.0, 0.0), (False, 1e-2, 0.1)):
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.SGD([ref], lr=0.1, momentum=0.9, weight_decay=wd, nesterov=nesterov, dampening=damp)
        p, m = p0.clone().to(DEV), torch.zeros(n, device=DEV)
        sh = torch.zeros(n, dtype=BF16, device=DEV)
        for i, g in enumerate(grads):
            ref.grad = g.clone()
            opt.step()
            gd = g.to(DEV)
            assert lib.tok_sgd_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), sh.data_ptr(), n, 0.1, 0.9, damp, wd,
                                    int(nesterov), int(i == 0), 0, st) == 0
        torch.cuda.synchronize()
        assert maxrel(p, ref.data, 1e-3) < 1e-4   # fma contraction differs from ATen's mul/add sequence
        assert torch.equal(sh.cpu(), p.cpu().to(BF16))
    for decoupled, cls in ((0, torch.optim.Adam), (1, torch.optim.AdamW)):
        ref = torch.nn.Parameter(p0.clone())
        opt = cls([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
        p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for i, g in enumerate(grads):
            ref.grad = g.clone()
            opt.step()
            gd = g.to(DEV)
            assert lib.tok_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), None, n, 1e-3, 0.9,
                                     0.999, 1e-8, 1e-2, decoupled, i + 1, 0, st) == 0
        torch.cuda.synchronize()
        assert maxrel(p, ref.data, 1e-3) < 1e-4


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, 3, 1, 1), (2, 16, 16, 256, 64, 1, 1, 0),
                                  (3, 14, 14, 64, 256, 1, 2, 0), (2, 17, 19, 64, 128, 3, 2, 1),
                                  (1, 7, 7, 512, 512, 3, 1, 1), (1, 24, 40, 96, 96, 3, 1, 1),
                                  (2, 16, 32, 192, 192, 3, 1, 1), (2, 20, 36, 48, 48, 3, 1, 1),
                                  (1, 16, 32, 384, 384, 3, 1, 1), (8, 128, 128, 48, 48, 3, 1, 1)])
@pytest.mark.parametrize('with_mask', [0, 1])
def test_conv_dgrad_bnstats(libs, case, with_mask):
    """dgrad whose epilogue also reduces sum(dz), sum(dz*y) of the unit that produced x."""
    n, h, w, c, k, r, stride, pad = case
    d = _desc(*case)
    lib, fake = libs
    dy = rnd(n, d.p, d.q, k).to(BF16)
    wd = rnd(c, r, r, k, scale=(r * r * k) ** -0.5).to(BF16)
    dx = rnd(n, h, w, c, seed=5).to(BF16)
    bn_y = rnd(n, h, w, c, seed=6).to(BF16)
    mask = torch.randint(0, 256, (n * h * w, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
    assert rows > 0
    part_d = torch.zeros(2, rows, c)
    part_h = torch.zeros(2, 2, c)
    dv = both(libs, 'tok_conv_dgrad_bnstats',
              lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d, f(dy), f(wd), f(dx), 1, f(bn_y),
                         f(mask) if with_mask else None, f(part_d) if f.__name__ == 'to_dev' else f(part_h), None])
    assert relerr(dv[id(dx)].float(), dx.float()) < 5e-3
    # statistics are taken over the fp32 dx (before the bf16 store): compare with rounding slack
    g = dv[id(dx)].float().cpu().reshape(-1, c)
    bits = ((mask.long().unsqueeze(-1) >> torch.arange(8)) & 1).reshape(-1, c).float() if with_mask else 1.0
    dz = g * bits
    got = dv[id(part_d)].cpu().sum(1)
    assert relerr(got[0], dz.sum(0)) < 4e-3
    assert relerr(got[1], (dz * bn_y.float().reshape(-1, c)).sum(0)) < 4e-3
    # and the dz*y form finalises to the same dgamma/dbeta/coefficients as the xhat form
    m = n * h * w
    mean, rstd, gamma = rnd(c, seed=8) * 0.3, rnd(c, seed=9).abs() + 0.5, rnd(c, seed=10) + 1
    xhat = (bn_y.float().reshape(-1, c) - mean) * rstd
    ph = torch.stack([dz.sum(0), (dz * xhat).sum(0)]).reshape(2, 1, c).contiguous()
    outs = []
    for partial, nrows, form in ((got.reshape(2, 1, c).contiguous(), 1, 1), (ph, 1, 0)):
        dg, db, coef = torch.zeros(c), torch.zeros(c), torch.zeros(3, c)
        assert fake.tok_bn_bwd_finalize(partial.data_ptr(), nrows, m, c, c, gamma.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), dg.data_ptr(), db.data_ptr(), coef.data_ptr(), 0, form,
                                        None) == 0
        outs.append((dg, db, coef))
    for a_, b_ in zip(outs[0], outs[1]):
        assert relerr(a_, b_) < 6e-3   # fp32-sum vs rounded-dx sums, amplified by the mean subtraction
    st = torch.cuda.current_stream().cuda_stream
    dev = lambda t: t.to(DEV)  # noqa: E731
    dgd, dbd, cod = dev(torch.zeros(c)), dev(torch.zeros(c)), dev(torch.zeros(3, c))
    pd, gd, md, rd = dev(got.reshape(2, 1, c).contiguous()), dev(gamma), dev(mean), dev(rstd)
    assert lib.tok_bn_bwd_finalize(pd.data_ptr(), 1, m, c, c, gd.data_ptr(), md.data_ptr(), rd.data_ptr(), dgd.data_ptr(),
                                   dbd.data_ptr(), cod.data_ptr(), 0, 1, st) == 0
    torch.cuda.synchronize()
    assert relerr(dgd, outs[0][0]) < 1e-3 and relerr(cod, outs[0][2]) < 1e-3


def test_tr_read_microbench(libs):
    """The wgrad kernel rests on ds_read_b64_tr_b16; a 1x1 conv wgrad with one-hot operands makes
    any lane-mapping mistake visible as a permuted dW."""
    n, h, w, c, k = 1, 8, 8, 64, 64
    d = _desc(n, h, w, c, k, 1, 1, 0)
    x = torch.zeros(n, h, w, c, dtype=BF16)
    dy = torch.zeros(n, h, w, k, dtype=BF16)
    flat_x, flat_dy = x.view(-1, c), dy.view(-1, k)
    for m in range(64):
        flat_x[m, (m * 7) % c] = 1.0 + m / 64.0
        flat_dy[m, (m * 5 + 3) % k] = 2.0
    dw = torch.zeros(k, 1, 1, c)
    wsb = libs[0].tok_conv_wgrad_ws_bytes(ctypes.byref(d))
    ws = torch.zeros(max(wsb // 4, 16))
    dv = both(libs, 'tok_conv_wgrad', lambda f: [ctypes.byref(d) if f.__name__ == 'to_dev' else d,
                                                 f(x), f(dy), f(dw), k, c, f(ws), wsb, 0, None])
    assert torch.allclose(dv[id(dw)].cpu(), dw, atol=1e-6)


# ---- metric-learning kernels (metric.hip) -------------------------------------------------------------------
@pytest.mark.parametrize('rows,c,ld,f32', [(37, 64, 64, 0), (16, 20, 24, 0), (1000, 512, 512, 0), (10, 64, 64, 1),
                                          (1003, 2048, 2048, 1)])
def test_l2norm(libs, rows, c, ld, f32):
    dt = torch.float32 if f32 else BF16
    x = torch.zeros(rows, ld)
    x[:, :c] = rnd(rows, c, scale=3.0)
    x = x.to(dt)
    y, inv = torch.empty(rows, ld, dtype=dt), torch.empty(rows)
    dv = both(libs, 'tok_l2norm_fwd', lambda d: [d(x), d(y), d(inv), rows, c, ld, f32, 1e-12, None])
    assert relerr(dv[id(inv)], inv) < 1e-5
    assert relerr(dv[id(y)].float(), y.float()) < (1e-6 if f32 else 4e-3)
    g = torch.zeros(rows, ld)
    g[:, :c] = rnd(rows, c, seed=3)
    g = g.to(dt)
    for acc in (0, 1):
        dx = (rnd(rows, ld, seed=9) if acc else torch.empty(rows, ld)).to(dt)
        if acc:
            dx[:, c:] = 0
        dv = both(libs, 'tok_l2norm_bwd', lambda d: [d(g), d(y), d(inv), d(dx), acc, rows, c, ld, f32, None])
        assert relerr(dv[id(dx)].float(), dx.float()) < (1e-5 if f32 else 6e-3)


@pytest.mark.parametrize('rows,classes,ld,easy', [(16, 10, 16, 0), (16, 10, 16, 1), (256, 1000, 1000, 0),
                                                  (33, 11003, 11008, 0)])
def test_arcface_margin(libs, rows, classes, ld, easy):
    import math
    m, scale = 0.5, 30.0
    consts = [math.cos(m), math.sin(m), math.cos(math.pi - m), math.sin(math.pi - m) * m, easy, scale]
    cs = torch.zeros(rows, ld)
    cs[:, :classes] = torch.rand(rows, classes, generator=torch.Generator().manual_seed(rows)) * 2 - 1
    tg = torch.randint(0, classes, (rows,), generator=torch.Generator().manual_seed(1))
    cs[0, tg[0]] = 1.0           # sine clamp edge
    cs[1, tg[1]] = -1.0
    cs[2, tg[2]] = 0.0
    cs = cs.to(BF16)
    out = torch.empty(rows, ld, dtype=BF16)
    dv = both(libs, 'tok_arcface_margin_fwd', lambda d: [d(cs), d(tg), rows, classes, ld, *consts, d(out), None])
    assert relerr(dv[id(out)].float(), out.float()) < 4e-3
    g = rnd(rows, ld, seed=4).to(BF16)
    dc = torch.empty(rows, ld, dtype=BF16)
    dv = both(libs, 'tok_arcface_margin_bwd', lambda d: [d(cs), d(tg), d(g), rows, classes, ld, *consts, d(dc), None])
    assert relerr(dv[id(dc)].float(), dc.float()) < 4e-3
    assert float(dv[id(dc)][:, classes:].abs().max().item() if ld > classes else 0.0) == 0.0


@pytest.mark.parametrize('na,nb', [(16, 16), (256, 256), (7, 130)])
def test_relevance_matrix_exact(libs, na, nb):
    a = torch.randint(0, 9, (na,), generator=torch.Generator().manual_seed(na))
    b = torch.randint(0, 9, (nb,), generator=torch.Generator().manual_seed(nb + 1))
    R = torch.empty(na, nb)
    dv = both(libs, 'tok_relevance_matrix', lambda d: [d(a), d(b), na, nb, d(R), None])
    assert torch.equal(dv[id(R)].cpu(), R)


@pytest.mark.parametrize('n1,n2,dim,ld,same', [(16, 16, 24, 24, 1), (16, 12, 24, 24, 0), (256, 256, 512, 512, 1),
                                               (33, 65, 20, 24, 0)])
def test_contrastive(libs, n1, n2, dim, ld, same):
    e1 = torch.zeros(n1, ld)
    e1[:, :dim] = rnd(n1, dim, scale=1.0 / dim ** 0.5)
    e1 = e1.to(BF16)
    if same:
        e2 = e1
    else:
        e2 = torch.zeros(n2, ld)
        e2[:, :dim] = rnd(n2, dim, scale=1.0 / dim ** 0.5, seed=5)
        e2 = e2.to(BF16)
    R = (torch.rand(n1, n2, generator=torch.Generator().manual_seed(2)) < 0.2).float()
    S, rl, loss = torch.empty(n1, n2), torch.empty(n1), torch.empty(1)
    dv = both(libs, 'tok_contrastive_fwd', lambda d: [d(e1), d(e2), d(R), n1, n2, dim, ld, 1.2, d(S), d(rl), d(loss),
                                                        None])
    assert relerr(dv[id(S)], S) < 1e-5 and relerr(dv[id(rl)], rl) < 1e-5 and relerr(dv[id(loss)], loss) < 1e-5
    gs = torch.tensor([0.7])
    de1 = torch.empty(n1, ld, dtype=BF16)
    de2 = torch.empty(n2, ld, dtype=BF16)
    dv = both(libs, 'tok_contrastive_bwd', lambda d: [d(e1), d(e1) if same else d(e2), d(R), d(S), d(gs), n1, n2, dim,
                                                        ld, 1.2, d(de1), None if same else d(de2), same, None])
    assert relerr(dv[id(de1)].float(), de1.float()) < 6e-3
    if not same:
        assert relerr(dv[id(de2)].float(), de2.float()) < 6e-3


@pytest.mark.parametrize('n,h,w,c', [(2, 8, 8, 16), (3, 9, 5, 64), (1, 1, 7, 8), (4, 56, 56, 256)])
def test_avgpool2x2(libs, n, h, w, c):
    x = rnd(n, h, w, c, seed=h).to(BF16)
    y = torch.empty(n, (h + 1) // 2, (w + 1) // 2, c, dtype=BF16)
    dv = both(libs, 'tok_avgpool2x2_fwd', lambda d: [d(x), d(y), n, h, w, c, None])
    assert relerr(dv[id(y)].float(), y.float()) < 4e-3
    ref = torch.nn.functional.avg_pool2d(x.float().permute(0, 3, 1, 2), 2, 2, ceil_mode=True, count_include_pad=False)
    assert relerr(dv[id(y)].float().cpu(), ref.permute(0, 2, 3, 1)) < 4e-3
    dy = rnd(*y.shape, seed=3).to(BF16)
    for acc in (0, 1):
        dx = rnd(n, h, w, c, seed=9).to(BF16)
        dv = both(libs, 'tok_avgpool2x2_bwd', lambda d: [d(dy), d(dx), acc, n, h, w, c, None])
        assert relerr(dv[id(dx)].float(), dx.float()) < 6e-3


@pytest.mark.parametrize('n,dim,ld,mode', [(16, 24, 24, 1), (33, 20, 24, 2), (512, 512, 512, 2), (7, 100, 104, 1)])
def test_embed_regulariser(libs, n, dim, ld, mode):
    e = torch.zeros(n, ld)
    e[:, :dim] = rnd(n, dim, seed=n)
    e[0] = 0                                   # a zero row: the L2 gradient is defined as 0 there
    e = e.to(BF16)
    rr, out = torch.empty(n), torch.empty(1)
    dv = both(libs, 'tok_embed_reg_fwd', lambda d: [d(e), n, dim, ld, mode, d(rr), d(out), None])
    assert relerr(dv[id(rr)], rr) < 1e-5 and relerr(dv[id(out)], out) < 1e-5
    gs, de = torch.tensor([0.3]), torch.empty(n, ld, dtype=BF16)
    dv = both(libs, 'tok_embed_reg_bwd', lambda d: [d(e), d(rr), d(gs), 1.0 / n, n, dim, ld, mode, d(de), None])
    assert relerr(dv[id(de)].float(), de.float()) < 6e-3
    assert float(dv[id(de)][0].abs().max()) == 0.0 and (ld == dim or float(dv[id(de)][:, dim:].abs().max()) == 0.0)


@pytest.mark.parametrize('rows,classes,ld,mean', [(37, 21, 24, 1), (4096, 80, 80, 1), (5, 1, 8, 0), (100000, 19, 24, 0)])
def test_bce_logits_ignore(libs, rows, classes, ld, mean):
    z = torch.zeros(rows, ld)
    z[:, :classes] = rnd(rows, classes, scale=3.0, seed=rows)
    z = z.to(BF16)
    g = torch.Generator().manual_seed(classes)
    tgt = (torch.rand(rows, classes, generator=g) < 0.3).float()
    tgt[torch.rand(rows, classes, generator=g) < 0.2] = -1
    loss = torch.empty(2050)
    dv = both(libs, 'tok_bce_logits_fwd', lambda d: [d(z), d(tgt), rows, classes, ld, -1.0, mean, d(loss), None])
    assert relerr(dv[id(loss)][:2], loss[:2]) < 2e-6 and float(dv[id(loss)][1]) == float((tgt != -1).sum())
    gs, dz = torch.tensor([1.3]), torch.empty(rows, ld, dtype=BF16)
    dv = both(libs, 'tok_bce_logits_bwd', lambda d: [d(z), d(tgt), d(loss), d(gs), rows, classes, ld, -1.0, mean, d(dz), None])
    assert relerr(dv[id(dz)].float(), dz.float()) < 6e-3
    assert ld == classes or float(dv[id(dz)][:, classes:].abs().max()) == 0.0


# ---- multi-resolution glue (resample.hip) ------------------------------------------------------------------
@pytest.mark.parametrize('n,h,w,c,shifts,relu', [(2, 16, 16, 16, (0, 1, 2, 3), 1), (2, 8, 24, 48, (0, 0, 1), 1),
                                               (1, 32, 32, 32, (0, 0), 1), (3, 8, 8, 64, (0, 2), 0)])
def test_fuse_sum_relu(libs, n, h, w, c, shifts, relu):
    terms = [rnd(n, h >> s, w >> s, c, seed=7 + i).to(BF16) for i, s in enumerate(shifts)]
    out = torch.empty(n, h, w, c, dtype=BF16)
    mask = torch.empty(n * h * w, c // 8, dtype=torch.uint8)

    def args(d):
        a = []
        for i in range(4):
            a += [d(terms[i]), shifts[i]] if i < len(terms) else [None, 0]
        return a + [n, h, w, c, relu, d(out), d(mask), None]
    dv = both(libs, 'tok_fuse_sum_relu_fwd', args)
    assert relerr(dv[id(out)].float(), out.float()) < 4e-3
    # mask bits may differ only where the sum rounds to +-0 in one of the two implementations
    assert (dv[id(mask)].cpu() != mask).float().mean() < 1e-3
    g = rnd(n, h, w, c, seed=3).to(BF16)
    for s in sorted(set(shifts)):
        for acc in (0, 1):
            dt = (rnd(n, h >> s, w >> s, c, seed=11) if acc else torch.zeros(n, h >> s, w >> s, c)).to(BF16)
            dv = both(libs, 'tok_fuse_sum_relu_bwd', lambda d: [d(g), d(mask) if relu else None, n, h, w, c, s, d(dt),
                                                                  acc, None])
            assert relerr(dv[id(dt)].float(), dt.float()) < 6e-3


@pytest.mark.parametrize('n,h,w,c,shifts,affine,relu', [(2, 16, 16, 16, (0, 1, 2, 3), (0, 1, 1, 1), 1), (2, 8, 24, 48, (0, 0, 1), (1, 0, 1), 1),
                                                      (1, 32, 32, 32, (0, 0), (1, 1), 1), (3, 8, 8, 64, (0, 2), (0, 0), 0)])
def test_fuse_sum_affine_relu(libs, n, h, w, c, shifts, affine, relu):
    """relu(sum_j (t_j * sc_j + sf_j | t_j)): terms handed over as raw conv outputs with their BatchNorm coefficients (the
    apply pass of a unit without activation folded into the HRNet fuse sum) beside terms taken as stored."""
    terms = [rnd(n, h >> s, w >> s, c, seed=7 + i).to(BF16) for i, s in enumerate(shifts)]
    scs = [rnd(c, seed=20 + i).abs() + 0.5 for i in range(len(shifts))]
    sfs = [rnd(c, seed=30 + i) for i in range(len(shifts))]
    out = torch.empty(n, h, w, c, dtype=BF16)
    mask = torch.empty(n * h * w, c // 8, dtype=torch.uint8)

    def args(d):
        a = []
        for i in range(4):
            if i < len(terms):
                a += [d(terms[i]), shifts[i]] + ([d(scs[i]), d(sfs[i])] if affine[i] else [None, None])
            else:
                a += [None, 0, None, None]
        return a + [n, h, w, c, relu, d(out), d(mask), None]
    dv = both(libs, 'tok_fuse_sum_affine_relu_fwd', args)
    assert relerr(dv[id(out)].float(), out.float()) < 4e-3
    assert (dv[id(mask)].cpu() != mask).float().mean() < 2e-3
    # against the definition, computed here
    acc = torch.zeros(n, h, w, c)
    for i, s_ in enumerate(shifts):
        v = terms[i].float()
        if affine[i]:
            v = v * scs[i] + sfs[i]
        if s_:
            v = v.repeat_interleave(1 << s_, dim=1).repeat_interleave(1 << s_, dim=2)
        acc = acc + v
    assert relerr(dv[id(out)].float(), acc.clamp_min(0) if relu else acc) < 4e-3


@pytest.mark.parametrize('n,hs,ws,c,hd,wd,ld,off', [(2, 8, 8, 16, 16, 16, 16, 0), (2, 4, 8, 32, 32, 64, 96, 32),
                                                    (1, 16, 16, 24, 64, 64, 24, 0), (2, 16, 16, 16, 16, 16, 48, 16),
                                                    (1, 5, 7, 8, 13, 20, 8, 0), (1, 9, 9, 8, 4, 5, 8, 0),
                                                    (2, 4, 4, 36, 16, 16, 272, 18), (1, 8, 8, 18, 8, 8, 56, 0)])
def test_bilinear(libs, n, hs, ws, c, hd, wd, ld, off):
    src = rnd(n, hs, ws, c).to(BF16)
    dst = rnd(n, hd, wd, ld, seed=2).to(BF16)       # other slices must stay untouched
    before = dst.clone()
    dv = both(libs, 'tok_bilinear_fwd', lambda d: [d(src), n, hs, ws, c, c, d(dst), hd, wd, ld, off, None])
    assert relerr(dv[id(dst)].float(), dst.float()) < 4e-3
    keep = torch.ones(ld, dtype=torch.bool)
    keep[off:off + c] = False
    assert torch.equal(dv[id(dst)].cpu()[..., keep], before[..., keep])
    g = rnd(n, hd, wd, ld, seed=5).to(BF16)
    for acc in (0, 1):
        ds = (rnd(n, hs, ws, c, seed=6) if acc else torch.zeros(n, hs, ws, c)).to(BF16)
        dv = both(libs, 'tok_bilinear_bwd', lambda d: [d(g), n, hd, wd, ld, off, d(ds), hs, ws, c, c, acc, None])
        assert relerr(dv[id(ds)].float(), ds.float()) < 6e-3


@pytest.mark.parametrize('n,h,w,c,lows', [(2, 16, 16, 16, [(8, 8), (4, 4), (2, 2)]), (1, 32, 64, 720, [(16, 32), (8, 16), (4, 8)]),
                                         (2, 13, 20, 24, [(5, 7)]), (1, 8, 8, 2056, [(4, 4), (2, 2)]), (3, 8, 8, 8, []),
                                         (2, 48, 32, 40, [(6, 4), (24, 16)]), (1, 32, 32, 24, [(8, 8)]), (1, 16, 32, 8, []),
                                         (1, 32, 32, 16, [(16, 16), (16, 16)]), (1, 32, 32, 16, [(11, 11)])])
def test_bilinear_sum_stats(libs, n, h, w, c, lows):
    """y = y0 + sum_j up(t_j), in place, + the BatchNorm partial rows of the rounded sum (the commuted HRNet neck)."""
    lib, fake = libs
    y0 = rnd(n, h, w, c).to(BF16)
    ts = [rnd(n, hs, ws, c, seed=3 + j).to(BF16) for j, (hs, ws) in enumerate(lows)]
    rows = lib.tok_bilinear_sum_stats_rows(n, h, w, c)
    assert rows >= 1
    st_dev = torch.zeros(2, rows, c, device=DEV)
    st_host = torch.zeros(2, fake.tok_bilinear_sum_stats_rows(n, h, w, c), c)
    yd = y0.to(DEV)
    td = [t.to(DEV) for t in ts]

    def args(y, tt, stats):
        a = [y.data_ptr()]
        for j in range(3):
            a += [tt[j].data_ptr(), lows[j][0], lows[j][1]] if j < len(tt) else [None, 1, 1]
        return a + [n, h, w, c, y.data_ptr(), stats.data_ptr()]
    assert lib.tok_bilinear_sum_stats(*args(yd, td, st_dev), torch.cuda.current_stream().cuda_stream) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    yh = y0.clone()
    assert fake.tok_bilinear_sum_stats(*args(yh, ts, st_host), None) == 0
    assert relerr(yd.float(), yh.float()) < 4e-3
    # the statistics are those of the ROUNDED values the kernel stored, folded over its partial rows
    f = yd.float().reshape(-1, c).double()
    got = st_dev.double().sum(1).cpu()
    assert torch.allclose(got[0], f.sum(0).cpu(), rtol=1e-4, atol=1e-3 * (n * h * w) ** 0.5)
    assert relerr(got[1], (f * f).sum(0).cpu()) < 1e-5
    # stats = NULL: the sum alone, same bits
    y2 = y0.to(DEV)
    a = args(y2, td, st_dev)
    a[-1] = None
    assert lib.tok_bilinear_sum_stats(*a, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y2, yd)


@pytest.mark.parametrize('n,hd,wd,c,lows', [(2, 32, 48, 16, [(16, 24), (8, 12), (4, 6)]), (1, 16, 16, 720, [(8, 8), (4, 4), (2, 2)]),
                                           (2, 32, 32, 24, [(8, 8)]), (1, 64, 32, 72, [(8, 4), (32, 16)]),
                                           (1, 20, 20, 8, [(10, 10), (5, 5)]), (1, 32, 32, 8, [(11, 11)])])
def test_bilinear_bwd_multi(libs, n, hd, wd, c, lows):
    """up_j^T g for several sources in one pass over g == tok_bilinear_bwd once per source (tiled LDS form where every factor
    is 2 / 4 / 8 and the map is a multiple of 16, the gather launches otherwise) — and TOK-independent of which form ran."""
    lib, fake = libs
    g = rnd(n, hd, wd, c, seed=7).to(BF16)
    gd = g.to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    outs = [torch.full((n, hs, ws, c), 7.0, dtype=BF16, device=DEV) for hs, ws in lows]
    a = [gd.data_ptr(), n, hd, wd, c]
    for j in range(3):
        a += [outs[j].data_ptr(), lows[j][0], lows[j][1]] if j < len(lows) else [None, 1, 1]
    assert lib.tok_bilinear_bwd_multi(*a, st) == 0, lib.tok_last_error()
    for (hs, ws), o in zip(lows, outs):
        ref = torch.empty((n, hs, ws, c), dtype=BF16, device=DEV)
        assert lib.tok_bilinear_bwd(gd.data_ptr(), n, hd, wd, c, 0, ref.data_ptr(), hs, ws, c, c, 0, st) == 0
        torch.cuda.synchronize()
        # same taps and weights, another fp32 summation order: equal to bf16 rounding of the result
        assert relerr(o.float(), ref.float()) < 3e-3, (hs, ws)
        host = torch.zeros(n, hs, ws, c, dtype=BF16)
        assert fake.tok_bilinear_bwd(g.data_ptr(), n, hd, wd, c, 0, host.data_ptr(), hs, ws, c, c, 0, None) == 0
        assert relerr(o.float(), host.float()) < 6e-3, (hs, ws)


def test_bilinear_adjoint_property(libs):
    """<up(x), g> == <x, up^T(g)> at a size no oracle run is needed for (size-independent property)."""
    lib, _ = libs
    n, hs, ws, c, hd, wd = 2, 32, 64, 64, 128, 256
    x = rnd(n, hs, ws, c).to(BF16).cuda()
    g = rnd(n, hd, wd, c, seed=1).to(BF16).cuda()
    up = torch.empty(n, hd, wd, c, dtype=BF16, device='cuda')
    gx = torch.empty(n, hs, ws, c, dtype=BF16, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tok_bilinear_fwd(x.data_ptr(), n, hs, ws, c, c, up.data_ptr(), hd, wd, c, 0, st) == 0
    assert lib.tok_bilinear_bwd(g.data_ptr(), n, hd, wd, c, 0, gx.data_ptr(), hs, ws, c, c, 0, st) == 0
    a = float((up.double() * g.double()).sum())
    b = float((x.double() * gx.double()).sum())
    assert abs(a - b) < 2e-2 * (up.double().norm() * g.double().norm()).item() ** 0.5 + 1e-2 * abs(a)


def test_pack_weights_batched_equals_single_packs(libs):
    """tok_pack_weights_batched over a table of ragged weights == tok_pack_weight_both per weight (bit-exact)."""
    lib, _ = libs
    shapes = [(64, 7, 7, 3, 64, 8, 4), (64, 1, 1, 64, 64, 1, 64), (128, 3, 3, 64, 128, 3, 64), (19, 1, 1, 720, 24, 1, 720),
              (1000, 1, 1, 2048, 1000, 1, 2048), (48, 3, 3, 48, 48, 3, 48)]
    st = torch.cuda.current_stream().cuda_stream
    items, keep, block = [], [], 0
    for i, (k, r, s, c, kp, sp, cp) in enumerate(shapes):
        w = rnd(k, r, s, c, seed=i).cuda()
        f1 = torch.empty(kp * r * sp * cp, dtype=BF16, device='cuda')
        d1 = torch.empty(cp * r * s * kp, dtype=BF16, device='cuda')
        want_d = i != 0
        f2, d2 = torch.full_like(f1, 7.0), torch.full_like(d1, 7.0)
        assert lib.tok_pack_weight_both(w.data_ptr(), k, r, s, c, f1.data_ptr(), kp, sp, cp, d1.data_ptr(), st) == 0
        it = _C.PackItem(w.data_ptr(), f2.data_ptr(), d2.data_ptr() if want_d else None, k, r, s, c, kp, sp, cp, block)
        block += lib.tok_pack_item_blocks(ctypes.byref(it))
        items.append(it)
        keep.append((w, f1, d1, f2, d2, want_d))
    arr = (_C.PackItem * len(items))(*items)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    assert lib.tok_pack_weights_batched(table.data_ptr(), len(items), block, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    for w, f1, d1, f2, d2, want_d in keep:
        assert torch.equal(f1, f2)
        assert torch.equal(d1, d2) if want_d else bool((d2 == 7.0).all())


def test_bn_padded_channels(libs):
    """num_features (18) < padded activation width (24): parameter arrays are c_real long, padding channels get
    zero coefficients (finalize fwd / eval / bwd)."""
    m, c, cr, rows = 4096, 24, 18, 3
    stats = rnd(2, rows, c).abs() * m
    stats[1] += stats[0] ** 2 / m * 1.5
    gamma, beta, rm, rv = rnd(cr, seed=1) + 1.5, rnd(cr, seed=2), rnd(cr, seed=3), rnd(cr, seed=4).abs() + 0.5
    nbt = torch.zeros(1, dtype=torch.int64)
    mean, rstd, scale, shift = (torch.full((c,), 9.0) for _ in range(4))
    dv = both(libs, 'tok_bn_finalize', lambda f: [f(stats), rows, m, c, cr, f(gamma), f(beta), f(rm), f(rv), f(nbt), 0.1,
                                                  1e-5, f(mean), f(rstd), f(scale), f(shift), None])
    for t in (mean, rstd, scale, shift, rm, rv):
        assert relerr(dv[id(t)], t) < 1e-5
    assert float(dv[id(scale)][cr:].abs().max()) == 0.0 and float(dv[id(shift)][cr:].abs().max()) == 0.0
    sc2, sh2 = torch.full((c,), 9.0), torch.full((c,), 9.0)
    dv = both(libs, 'tok_bn_eval_coeffs', lambda f: [f(gamma), f(beta), f(rm), f(rv), 1e-5, c, cr, f(sc2), f(sh2), None])
    assert relerr(dv[id(sc2)], sc2) < 1e-6 and relerr(dv[id(sh2)], sh2) < 1e-6
    partial = rnd(2, rows, c, seed=8)
    dg, db, coef = torch.zeros(cr), torch.zeros(cr), torch.full((3, c), 9.0)
    dv = both(libs, 'tok_bn_bwd_finalize', lambda f: [f(partial), rows, m, c, cr, f(gamma), f(mean), f(rstd), f(dg), f(db),
                                                      f(coef), 0, 1, None])
    for t in (dg, db, coef):
        assert relerr(dv[id(t)], t) < 1e-5
    assert float(dv[id(coef)][:, cr:].abs().max()) == 0.0


# ---- token-major transformer kernels (transformer.hip) ---------------------------------------------------------
@pytest.mark.parametrize('rows,c,ld,sc,rs', [(37, 96, 96, 0, 0), (64, 768, 768, 1, 1), (50, 18, 24, 1, 0),
                                             (4096, 192, 192, 1, 1), (333, 384, 384, 1, 0), (130, 1024, 1024, 0, 1)])
def test_layernorm(libs, rows, c, ld, sc, rs):
    rps = rows // 2 if rows % 2 == 0 else rows
    x = torch.zeros(rows, ld)
    x[:, :c] = rnd(rows, c, scale=2.0) + 0.5
    x = x.to(BF16)
    short = torch.zeros(rows, ld)
    short[:, :c] = rnd(rows, c, seed=3)
    short = short.to(BF16)
    scale = torch.tensor([0.0, 2.0][:rows // rps]) if rs else None
    gamma, beta = rnd(c, seed=1) + 1.0, rnd(c, seed=2)
    out, mean, rstd = torch.empty(rows, ld, dtype=BF16), torch.empty(rows), torch.empty(rows)
    dv = both(libs, 'tok_layernorm_fwd', lambda d: [d(x), d(short) if sc else None, d(scale) if rs else None, rps,
                                                    d(gamma), d(beta), d(out), d(mean), d(rstd), rows, c, ld, 1e-5, None])
    assert relerr(dv[id(out)].float(), out.float()) < 4e-3
    assert relerr(dv[id(mean)], mean) < 1e-4 and relerr(dv[id(rstd)], rstd) < 1e-4
    g = torch.zeros(rows, ld)
    g[:, :c] = rnd(rows, c, seed=5)
    g = g.to(BF16)
    lib, fake = libs
    nrows = lib.tok_layernorm_bwd_rows(rows, c)
    for acc in (0, 1):
        dx_h = (rnd(rows, ld, seed=6) if acc else torch.zeros(rows, ld)).to(BF16)
        dx_h[:, c:] = 0
        dx_d = dx_h.cuda()
        part_h, part_d = torch.zeros(2, 1, c), torch.zeros(2, nrows, c, device='cuda')
        st = torch.cuda.current_stream().cuda_stream
        P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
        assert lib.tok_layernorm_bwd(P(g.cuda()), P(x.cuda()), P(mean.cuda()), P(rstd.cuda()), P(gamma.cuda()),
                                     P(scale.cuda()) if rs else None, rps, P(dx_d), acc, P(part_d), rows, c, ld, st) == 0
        assert fake.tok_layernorm_bwd(P(g), P(x), P(mean), P(rstd), P(gamma), P(scale) if rs else None, rps, P(dx_h), acc,
                                      P(part_h), rows, c, ld, None) == 0
        torch.cuda.synchronize()
        assert relerr(dx_d.float(), dx_h.float()) < 8e-3
        assert relerr(part_d.sum(1), part_h.sum(1)) < 2e-3
        red = torch.zeros(c, device='cuda') + 1.0
        assert lib.tok_colsum_f32(part_d[0].data_ptr(), nrows, c, red.data_ptr(), 1, st) == 0
        torch.cuda.synchronize()
        assert relerr(red - 1.0, part_d[0].double().sum(0)) < 1e-5


@pytest.mark.parametrize('kind', [0, 1])
def test_activation(libs, kind):
    x = (rnd(4096 * 8) * 2).to(BF16)
    y = torch.empty_like(x)
    dv = both(libs, 'tok_act_fwd', lambda d: [kind, d(x), d(y), x.numel(), None])
    assert relerr(dv[id(y)].float(), y.float()) < 4e-3
    g = rnd(4096 * 8, seed=2).to(BF16)
    for acc in (0, 1):
        dx = (rnd(4096 * 8, seed=3) if acc else torch.zeros(4096 * 8)).to(BF16)
        dv = both(libs, 'tok_act_bwd', lambda d: [kind, d(g), d(x), d(dx), acc, x.numel(), None])
        assert relerr(dv[id(dx)].float(), dx.float()) < 6e-3
    dx = rnd(4096 * 8, seed=4).to(BF16)
    dv = both(libs, 'tok_act_bwd', lambda d: [2, d(g), d(g), d(dx), 1, x.numel(), None])     # identity accumulate
    assert relerr(dv[id(dx)].float(), dx.float()) < 4e-3


@pytest.mark.parametrize('b,h,w,heads,ws,shift', [(2, 8, 8, 3, 4, 0), (2, 8, 8, 3, 4, 2), (1, 16, 16, 2, 8, 4),
                                                  (3, 7, 7, 6, 7, 0), (1, 32, 32, 1, 16, 8), (2, 14, 14, 3, 7, 3)])
def test_window_attention(libs, b, h, w, heads, ws, shift):
    lib, fake = libs
    c = heads * 32
    n, nw = ws * ws, (h // ws) * (w // ws)
    qkv = rnd(b * h * w, 3 * c).to(BF16)
    ls = torch.full((heads,), 2.3) + rnd(heads, seed=1) * 0.3
    ls[0] = 5.0                                            # above ln(100): the clamp is active for head 0
    bias = rnd(heads, n, n, seed=2)
    mask = None
    if shift:
        img = torch.zeros(1, h, w, 1)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, h // ws, ws, w // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, n)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0).contiguous()
    out, lse = torch.empty(b * h * w, c, dtype=BF16), torch.empty(b * nw * heads * n)
    dv = both(libs, 'tok_window_attn_fwd', lambda d: [d(qkv), b, h, w, c, heads, ws, shift, 3 * c, d(ls), d(bias),
                                                      d(mask) if mask is not None else None, d(out), d(lse), None])
    assert relerr(dv[id(out)].float(), out.float()) < 1e-2     # MFMA path: qn, kn, P are bf16 operands
    # lse layout: fake (B*nW, heads, N) == kernel ((b*nW + win)*heads + h, N)
    assert relerr(dv[id(lse)], lse) < 3e-3      # logits from bf16-rounded qn, kn on the MFMA path
    g = rnd(b * h * w, c, seed=7).to(BF16)
    rows = lib.tok_window_attn_bwd_rows(b, h, w, heads, ws)
    assert rows <= b * nw
    # host reference with its own lse; the device backward gets the lse of the DEVICE forward (as in training: the
    # recomputed probabilities must be normalised by the same bf16-operand logits that produced them)
    P = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    dq_h = torch.empty(b * h * w, 3 * c, dtype=BF16)
    scr_h, dsp_h = torch.zeros(b * nw, heads * n * n), torch.zeros(b * nw, heads)
    assert fake.tok_window_attn_bwd(P(qkv), P(g), b, h, w, c, heads, ws, shift, 3 * c, P(ls), P(bias), P(mask), P(lse),
                                    P(dq_h), P(scr_h), P(dsp_h), None) == 0
    dq_d = torch.empty(b * h * w, 3 * c, dtype=BF16, device='cuda')
    scr_d, dsp_d = torch.zeros(rows, heads * n * n, device='cuda'), torch.zeros(rows, heads, device='cuda')
    dev_in = [t_.cuda() if t_ is not None else None for t_ in (qkv, g, ls, bias, mask)]
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tok_window_attn_bwd(P(dev_in[0]), P(dev_in[1]), b, h, w, c, heads, ws, shift, 3 * c, P(dev_in[2]),
                                   P(dev_in[3]), P(dev_in[4]), P(dv[id(lse)]), P(dq_d), P(scr_d), P(dsp_d), st) == 0, \
        lib.tok_last_error()
    torch.cuda.synchronize()
    assert relerr(dq_d.float(), dq_h.float()) < 3e-2
    assert relerr(scr_d.sum(0), scr_h.sum(0)) < 3e-2                     # d(bias)
    assert relerr(dsp_d.sum(0)[1:], dsp_h.sum(0)[1:]) < 3e-2 or heads == 1   # d(logit_scale)
    assert float(dsp_d[:, 0].abs().max()) == 0.0                         # clamped head: zero gradient


@pytest.mark.parametrize('b,h,w,heads,ws,shift', [(256, 14, 14, 12, 7, 3), (256, 14, 14, 12, 7, 0), (128, 28, 28, 6, 7, 3),
                                                  (64, 56, 56, 3, 7, 3), (256, 7, 7, 24, 7, 0)])
def test_window_attention_is_bit_reproducible(libs, b, h, w, heads, ws, shift):
    """The SwinV2-T stage geometries at sizes where a workgroup walks several images (units >= 3072): the same launch, six
    times, gives the same bits in d(qkv), the d(logits) partial rows and the d(logit_scale) partials, and every element is
    written.  Round 3 found a few hundred wrong d(q) / d(k) elements per launch, different ones every run (a register reused
    under two ds_bpermute in flight, transformer.hip: the value barrier after the delta reduction); only the full-size model
    test saw it."""
    lib, _ = libs
    c, n, nw = heads * 32, ws * ws, (h // ws) * (w // ws)
    gen = torch.Generator(device='cuda').manual_seed(1)
    qkv = torch.randn(b * h * w, 3 * c, device='cuda', generator=gen).to(BF16)
    g = torch.randn(b * h * w, c, device='cuda', generator=gen).to(BF16)
    ls = torch.full((heads,), 2.3, device='cuda')
    bias = torch.randn(heads, n, n, device='cuda', generator=gen)
    mask = None
    if shift:
        img = torch.zeros(1, h, w, 1)
        cnt = 0
        for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
                img[:, hs, wsl, :] = cnt
                cnt += 1
        mw = img.view(1, h // ws, ws, w // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, n)
        am = mw.unsqueeze(1) - mw.unsqueeze(2)
        mask = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0).contiguous().cuda()
    P = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    fw = []
    for _ in range(3):
        out = torch.full((b * h * w, c), float('nan'), dtype=BF16, device='cuda')
        lse = torch.full((b * nw * heads * n,), float('nan'), device='cuda')
        assert lib.tok_window_attn_fwd(P(qkv), b, h, w, c, heads, ws, shift, 3 * c, P(ls), P(bias), P(mask), P(out), P(lse),
                                       st) == 0, lib.tok_last_error()
        fw.append((out, lse))
    torch.cuda.synchronize()
    assert not fw[0][0].isnan().any() and not fw[0][1].isnan().any()
    assert all(torch.equal(fw[0][0], o) and torch.equal(fw[0][1], l_) for o, l_ in fw[1:])
    rows = lib.tok_window_attn_bwd_rows(b, h, w, heads, ws)
    res = []
    for _ in range(6):
        dq = torch.full((b * h * w, 3 * c), float('nan'), dtype=BF16, device='cuda')
        scr = torch.full((rows, heads * n * n), float('nan'), device='cuda')
        dsp = torch.full((rows, heads), float('nan'), device='cuda')
        assert lib.tok_window_attn_bwd(P(qkv), P(g), b, h, w, c, heads, ws, shift, 3 * c, P(ls), P(bias), P(mask), P(fw[0][1]),
                                       P(dq), P(scr), P(dsp), st) == 0, lib.tok_last_error()
        res.append((dq, scr, dsp))
    torch.cuda.synchronize()
    assert not any(t_.isnan().any() for t_ in res[0])
    for r_ in res[1:]:
        for a_, b_ in zip(res[0], r_):
            assert torch.equal(a_, b_), int((a_ != b_).sum())


def test_cpb_bias_and_patch_merge(libs):
    heads, ws = 3, 4
    n, T_ = ws * ws, (2 * ws - 1) ** 2
    coords = torch.flatten(torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij')), 1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    index = rel.sum(-1).contiguous()
    table = rnd(T_, 8).to(BF16)
    bias = torch.empty(heads, n, n)
    dv = both(libs, 'tok_cpb_bias_fwd', lambda d: [d(table), 8, d(index), heads, n, d(bias), None])
    assert relerr(dv[id(bias)], bias) < 1e-5
    for tr in (0, 1):
        db = rnd(heads, n, n, seed=4)
        dt = torch.full((T_, 8), float('nan'), dtype=BF16)
        dv = both(libs, 'tok_cpb_bias_bwd', lambda d: [d(db), tr, d(table), 8, d(index), heads, n, T_, d(dt), None])
        assert relerr(dv[id(dt)].float(), dt.float()) < 6e-3          # incl. zeroed padding columns
    x = rnd(2, 8, 12, 16).to(BF16)
    y = torch.empty(2, 4, 6, 64, dtype=BF16)
    dv = both(libs, 'tok_patch_merge', lambda d: [d(x), d(y), 2, 8, 12, 16, 0, None])
    assert torch.equal(dv[id(y)].cpu(), y)                       # permutation: bit-exact
    back = torch.empty_like(x)
    dv2 = both(libs, 'tok_patch_merge', lambda d: [d(y), d(back), 2, 8, 12, 16, 1, None])
    assert torch.equal(dv2[id(back)].cpu(), x)                   # inverse round trip


@pytest.mark.parametrize('m,n', [(200704, 288), (1000, 8), (50176, 3072), (77, 768)])
def test_colsum_partial(libs, m, n):
    lib, _ = libs
    x = rnd(m, n).to(BF16).cuda()
    rows = lib.tok_colsum_partial_rows(m, n)
    part = torch.empty(rows, n, device='cuda')
    out = torch.empty(n, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    assert lib.tok_colsum_partial(x.data_ptr(), m, n, part.data_ptr(), st) == 0, lib.tok_last_error()
    assert lib.tok_colsum_f32(part.data_ptr(), rows, n, out.data_ptr(), 0, st) == 0
    torch.cuda.synchronize()
    ref = x.double().sum(0)
    assert float((out.double() - ref).abs().max()) < 1e-4 * float(x.double().abs().sum(0).max())
    # the pair launch gives the bits of two single launches (second one accumulating)
    part2 = torch.randn(rows, n, device='cuda')
    base = torch.randn(n, device='cuda')
    a0, a1 = torch.empty(n, device='cuda'), base.clone()
    b0, b1 = torch.empty(n, device='cuda'), base.clone()
    assert lib.tok_colsum_f32(part.data_ptr(), rows, n, a0.data_ptr(), 0, st) == 0
    assert lib.tok_colsum_f32(part2.data_ptr(), rows, n, a1.data_ptr(), 1, st) == 0
    assert lib.tok_colsum_f32_pair(part.data_ptr(), part2.data_ptr(), rows, n, b0.data_ptr(), 0, b1.data_ptr(), 1, st) == 0, \
        lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


@pytest.mark.parametrize('rows,cols', [(1024, 7203), (256, 14406), (64, 57624), (1, 2048), (37, 2049), (1000, 4099)])
@pytest.mark.parametrize('acc', [0, 1])
def test_colsum_f32_wide_matrices(libs, rows, cols, acc):
    """The fold of window attention's d(bias) partial rows (heads x 49 x 49 columns: odd for three heads) on the wide kernel
    (cols >= 2048, csrc/transformer.hip: colsum_f32_wide_kernel): column sums in fp64 order-of-rows, accumulate, run twice
    bit-identical, nothing written past the last column."""
    lib, _ = libs
    g = torch.Generator(device='cuda').manual_seed(rows + cols)
    src = torch.randn(rows, cols, device='cuda', generator=g)
    base = torch.randn(cols + 8, device='cuda', generator=g)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for _ in range(2):
        out = base.clone()
        assert lib.tok_colsum_f32(src.data_ptr(), rows, cols, out.data_ptr(), acc, st) == 0, lib.tok_last_error()
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0][cols:], base[cols:])
    ref = src.double().sum(0) + (base[:cols].double() if acc else 0)
    assert float((outs[0][:cols].double() - ref).abs().max()) < 2e-6 * float(src.double().abs().sum(0).max() + 1)


@pytest.mark.parametrize('case', [(4, 16, 16, 64, 64, 3, 1, 1), (2, 14, 14, 256, 512, 1, 1, 0), (3, 9, 11, 32, 24, 3, 2, 1),
                                  (8, 28, 28, 128, 128, 3, 1, 1)])
def test_fused_bn_finalize_equals_standalone(libs, case):
    """tok_conv_fwd_bn / tok_conv_dgrad_bn ("last workgroup of a channel tile finalizes") == tok_conv_fwd + tok_bn_finalize
    / tok_conv_dgrad_bnstats + tok_bn_bwd_finalize, and the ticket counters come back zero."""
    lib, _ = libs
    n, h, w, c, k, r, stride, pad = case
    p, q = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    d = _C.ConvDesc(n, h, w, c, k, r, r, p, q, stride, pad, r)
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    dev = 'cuda'
    x = rnd(n, h, w, c).to(BF16).to(dev)
    wt = rnd(k, r, r, c, seed=1, scale=0.1).to(dev)
    wf = torch.empty(k, r, r, c, dtype=BF16, device=dev)
    wd = torch.empty(c, r, r, k, dtype=BF16, device=dev)
    assert lib.tok_pack_weight_both(P(wt), k, r, r, c, P(wf), k, r, c, P(wd), st) == 0
    k_real = k - 3 if k % 8 == 0 and k > 8 else k
    gamma, beta = (rnd(k_real, seed=2) + 1.0).to(dev), rnd(k_real, seed=3).to(dev)
    m = n * p * q
    rows = lib.tok_conv_fwd_stat_rows(ctypes.byref(d))
    counters = torch.zeros(64, dtype=torch.int32, device=dev)
    res = []
    for fused in (0, 1):
        y = torch.empty(n, p, q, k, dtype=BF16, device=dev)
        stats = torch.empty(2, rows, k, device=dev)
        rm, rv = torch.zeros(k_real, device=dev), torch.ones(k_real, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        mean, rstd, scale, shift = (torch.full((k,), 7.0, device=dev) for _ in range(4))
        if fused:
            fb = _C.BnFused(P(counters), m, k_real, 0, 0.1, 1e-5, P(gamma), P(beta), P(rm), P(rv), P(nbt), P(mean), P(rstd),
                            P(scale), P(shift), None, None, None)
            assert lib.tok_conv_fwd_bn(ctypes.byref(d), P(x), P(wf), P(y), P(stats), ctypes.byref(fb), st) == 0, \
                lib.tok_last_error()
        else:
            assert lib.tok_conv_fwd(ctypes.byref(d), P(x), P(wf), None, P(y), P(stats), st) == 0
            assert lib.tok_bn_finalize(P(stats), rows, m, k, k_real, P(gamma), P(beta), P(rm), P(rv), P(nbt), 0.1, 1e-5,
                                       P(mean), P(rstd), P(scale), P(shift), st) == 0
        torch.cuda.synchronize()
        res.append((y, mean, rstd, scale, shift, rm, rv, nbt))
    assert int(counters.abs().sum()) == 0
    assert int(res[1][7]) == 1
    # same kernel on both paths -> same bits; where the plain launch rides the shared-window kernel (conv_win.hip sums the
    # reduction chunk-major, the two-buffer kernel of the fused entry point tap-major) the outputs agree to bf16 rounding
    same = torch.equal(res[0][0], res[1][0])
    assert same or relerr(res[1][0].float(), res[0][0].float()) < 3e-3
    for a, b in zip(res[0][1:7], res[1][1:7]):
        assert relerr(b, a) < (1e-6 if same else 2e-3)
    # backward: dgrad of this conv completing d(x) for a producer BatchNorm over x's channels
    if c % 8 == 0:
        y, mean_k = res[0][0], None
        dy = rnd(n, p, q, k, seed=5).to(BF16).to(dev)
        bn_y = rnd(n, h, w, c, seed=6).to(BF16).to(dev)
        mask = torch.randint(0, 256, (n * h * w, c // 8), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
        pmean, prstd = rnd(c, seed=7).to(dev), (rnd(c, seed=8).abs() + 0.5).to(dev)
        pgamma = (rnd(c, seed=9) + 1.0).to(dev)
        prow = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
        out = []
        for fused in (0, 1):
            dx = torch.empty(n, h, w, c, dtype=BF16, device=dev)
            part = torch.empty(2, prow, c, device=dev)
            dg, db, coef = torch.ones(c, device=dev), torch.ones(c, device=dev), torch.empty(3, c, device=dev)
            if fused:
                fb = _C.BnFused(P(counters), n * h * w, c, 1, 0.0, 0.0, P(pgamma), None, None, None, None, P(pmean),
                                P(prstd), None, None, P(dg), P(db), P(coef))
                assert lib.tok_conv_dgrad_bn(ctypes.byref(d), P(dy), P(wd), P(dx), 0, P(bn_y), P(mask), P(part),
                                             ctypes.byref(fb), st) == 0, lib.tok_last_error()
            else:
                assert lib.tok_conv_dgrad_bnstats(ctypes.byref(d), P(dy), P(wd), P(dx), 0, P(bn_y), P(mask), P(part), st) == 0
                assert lib.tok_bn_bwd_finalize(P(part), prow, n * h * w, c, c, P(pgamma), P(pmean), P(prstd), P(dg), P(db),
                                               P(coef), 1, 1, st) == 0
            torch.cuda.synchronize()
            out.append((dx, dg, db, coef))
        assert int(counters.abs().sum()) == 0
        same = torch.equal(out[0][0], out[1][0])
        assert same or relerr(out[1][0].float(), out[0][0].float()) < 3e-3
        for a, b in zip(out[0][1:], out[1][1:]):
            assert relerr(b, a) < (1e-5 if same else 5e-3)


@pytest.mark.parametrize('rows,classes,ld,mode,log_loss,sel', [(5000, 19, 24, 0, 0, None), (777, 5, 8, 0, 1, [0, 3]),
                                                              (4099, 1, 8, 1, 0, None), (100000, 3, 8, 0, 0, None),
                                                              (3001, 19, 24, 2, 0, None), (640, 6, 8, 2, 1, [1, 5])])
def test_dice(libs, rows, classes, ld, mode, log_loss, sel):
    lib, fake = libs
    z = (rnd(rows, ld) * 2).to(BF16)
    if mode == 0:
        t = torch.randint(0, max(classes - 1, 1), (rows,), generator=torch.Generator().manual_seed(2))   # last class empty
    elif mode == 2:
        t = (torch.rand(rows, classes, generator=torch.Generator().manual_seed(2)) < 0.3).float()
        t[:, classes - 1] = 0                                                                             # last class empty
    else:
        t = (torch.rand(rows, generator=torch.Generator().manual_seed(2)) < 0.3).float()
    selt = torch.tensor(sel) if sel else None
    nparts = lib.tok_dice_rows(rows)
    P = lambda x: None if x is None else x.data_ptr()   # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    part_h, loss_h, coef_h = torch.zeros(1, 3, classes), torch.zeros(1), torch.zeros(2, classes)
    assert fake.tok_dice_fwd(P(z), P(t), rows, classes, ld, mode, 0.5, 1e-7, log_loss, P(selt), len(sel) if sel else 0,
                             P(part_h), P(loss_h), P(coef_h), None) == 0
    zd, td = z.cuda(), t.cuda()
    seld = selt.cuda() if sel else None
    part_d, loss_d, coef_d = torch.zeros(nparts, 3, classes, device='cuda'), torch.zeros(1, device='cuda'), \
        torch.zeros(2, classes, device='cuda')
    assert lib.tok_dice_fwd(P(zd), P(td), rows, classes, ld, mode, 0.5, 1e-7, log_loss, P(seld), len(sel) if sel else 0,
                            P(part_d), P(loss_d), P(coef_d), st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert relerr(loss_d, loss_h) < 1e-4 and relerr(coef_d, coef_h) < 1e-3
    gs = torch.tensor([1.7])
    d_h = torch.empty(rows, ld, dtype=BF16)
    d_d = torch.full((rows, ld), float('nan'), dtype=BF16, device='cuda')
    assert fake.tok_dice_bwd(P(z), P(t), P(coef_h), P(gs), rows, classes, ld, mode, P(d_h), None) == 0
    assert lib.tok_dice_bwd(P(zd), P(td), P(coef_d), P(gs.cuda()), rows, classes, ld, mode, P(d_d), st) == 0
    torch.cuda.synchronize()
    assert relerr(d_d.float(), d_h.float()) < 6e-3


@pytest.mark.parametrize('n,d,ld', [(24, 40, 40), (512, 128, 128), (64, 20, 24)])
def test_ntxent_kernels(libs, n, d, ld):
    e = torch.zeros(n, ld)
    e[:, :d] = torch.nn.functional.normalize(rnd(n, d), dim=1)
    e = e.to(BF16)
    lse, rl, loss = torch.empty(n), torch.empty(n), torch.empty(1)
    dv = both(libs, 'tok_ntxent_fwd', lambda f: [f(e), n, d, ld, 0.2, f(lse), f(rl), f(loss), None])
    assert relerr(dv[id(lse)], lse) < 1e-5 and relerr(dv[id(loss)], loss) < 1e-5
    gs = torch.tensor([1.3])
    de = torch.full((n, ld), float('nan'), dtype=BF16)
    dv = both(libs, 'tok_ntxent_bwd', lambda f: [f(e), f(lse), f(gs), n, d, ld, 0.2, f(de), None])
    assert relerr(dv[id(de)].float(), de.float()) < 6e-3


@pytest.mark.parametrize('rows,d,ld,swap', [(16, 24, 24, 0), (1000, 128, 128, 1), (33, 20, 24, 1)])
def test_triplet_kernels(libs, rows, d, ld, swap):
    ts = []
    for i in range(3):
        t = torch.zeros(rows, ld)
        t[:, :d] = rnd(rows, d, seed=i, scale=0.7)
        ts.append(t.to(BF16))
    a, p, ng = ts
    dist, rl, loss = torch.empty(rows, 3), torch.empty(rows), torch.empty(1)
    dv = both(libs, 'tok_triplet_fwd', lambda f: [f(a), f(p), f(ng), rows, d, ld, 0.6, 1e-6, swap, f(dist), f(rl), f(loss), None])
    assert relerr(dv[id(dist)], dist) < 1e-5 and relerr(dv[id(loss)], loss) < 1e-5
    gs = torch.tensor([0.9])
    outs = [torch.full((rows, ld), float('nan'), dtype=BF16) for _ in range(3)]
    dv = both(libs, 'tok_triplet_bwd', lambda f: [f(a), f(p), f(ng), f(dist), f(gs), rows, d, ld, 0.6, 1e-6, swap, f(outs[0]),
                                                  f(outs[1]), f(outs[2]), None])
    for o in outs:
        assert relerr(dv[id(o)].float(), o.float()) < 6e-3


# ---- DaViT ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('b,h,w,heads,ws', [(2, 8, 8, 3, 4), (3, 14, 14, 6, 7), (9, 7, 7, 2, 7)])
def test_window_attention_plain_mode(libs, b, h, w, heads, ws):
    """logit_scale == bias == NULL: softmax(q k^T / sqrt(32)) v of DaViT's WindowAttention (davit.py:168-207)."""
    lib, fake = libs
    c = heads * 32
    n, nw = ws * ws, (h // ws) * (w // ws)
    qkv = rnd(b * h * w, 3 * c).to(BF16)
    out, lse = torch.empty(b * h * w, c, dtype=BF16), torch.empty(b * nw * heads * n)
    dv = both(libs, 'tok_window_attn_fwd', lambda d: [d(qkv), b, h, w, c, heads, ws, 0, 3 * c, None, None, None, d(out),
                                                      d(lse), None])
    x = qkv.float().view(b, h // ws, ws, w // ws, ws, 3, heads, 32).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, heads, n, 32)
    ref = torch.softmax(x[0] @ x[1].transpose(-1, -2) / 32 ** 0.5, -1) @ x[2]                  # torch fp32 restatement
    ref = ref.view(b, h // ws, w // ws, heads, ws, ws, 32).permute(0, 1, 4, 2, 5, 3, 6).reshape(b * h * w, c)
    assert relerr(out.float(), ref) < 1e-2 and relerr(dv[id(out)].float(), ref) < 1e-2
    assert relerr(dv[id(lse)], lse) < 3e-3
    g = rnd(b * h * w, c, seed=7).to(BF16)
    P = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    dq_h = torch.empty(b * h * w, 3 * c, dtype=BF16)
    assert fake.tok_window_attn_bwd(P(qkv), P(g), b, h, w, c, heads, ws, 0, 3 * c, None, None, None, P(lse), P(dq_h), None,
                                    None, None) == 0
    dq_d = torch.empty(b * h * w, 3 * c, dtype=BF16, device='cuda')
    qd, gd = qkv.cuda(), g.cuda()
    assert lib.tok_window_attn_bwd(P(qd), P(gd), b, h, w, c, heads, ws, 0, 3 * c, None, None, None, P(dv[id(lse)]), P(dq_d),
                                   None, None, torch.cuda.current_stream().cuda_stream) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert relerr(dq_d.float(), dq_h.float()) < 3e-2
    ls = torch.zeros(heads, device='cuda')
    assert lib.tok_window_attn_fwd(P(qd), b, h, w, c, heads, ws, 0, 3 * c, P(ls), None, None, P(dv[id(out)]), P(dv[id(lse)]),
                                   None) != 0                                              # scale without bias: refused


@pytest.mark.parametrize('images,rpi,heads,pad', [(2, 100, 3, 0), (3, 49, 6, 8), (1, 3136, 3, 0), (5, 1, 2, 0)])
def test_channel_attention_kernels(libs, images, rpi, heads, pad):
    c = heads * 32
    ld = 3 * c + pad
    n = images * rpi
    qkv = rnd(n, ld).to(BF16)
    P = lambda t_, off=0: t_.data_ptr() + 2 * off   # noqa: E731
    scale = 32 ** -0.5
    for mode in (0, 1):
        a = torch.empty(images * heads, 32, 32)
        dv = both(libs, 'tok_chan_gram', lambda d: (lambda q_: [q_.data_ptr() + 2 * c, ld, q_.data_ptr() + 4 * c, ld, rpi,
                                                                images, heads, scale, mode, None, d(a), None])(d(qkv)))
        k = qkv[:, c:2 * c].float().view(images, rpi, heads, 32).permute(0, 2, 1, 3)
        v = qkv[:, 2 * c:3 * c].float().view(images, rpi, heads, 32).permute(0, 2, 1, 3)
        ref = (scale * k.transpose(-1, -2) @ v).reshape(-1, 32, 32)                       # torch fp32 restatement
        ref = ref.softmax(-1) if mode else ref
        assert relerr(a, ref) < 1e-5 and relerr(dv[id(a)], ref) < 1e-4
    attn = a
    gm = torch.empty_like(attn)
    g = rnd(n, c, seed=3).to(BF16)
    dv = both(libs, 'tok_chan_gram', lambda d: [d(g), c, d(qkv), ld, rpi, images, heads, 1.0, 2, d(attn), d(gm), None])
    assert relerr(dv[id(gm)], gm) < 1e-4
    for transposed in (0, 1):
        out = torch.zeros(n, ld, dtype=BF16)
        dv = both(libs, 'tok_chan_apply', lambda d: [d(qkv), ld, d(attn), transposed, 0.5, rpi, images, heads,
                                                     d(out).data_ptr() + 2 * c, ld, None])
        q = qkv[:, :c].float().view(images, rpi, heads, 32).permute(0, 2, 1, 3)
        m = attn.view(images, heads, 32, 32)
        ref = 0.5 * (q @ (m if transposed else m.transpose(-1, -2)))
        ref = ref.permute(0, 2, 1, 3).reshape(n, c)
        assert relerr(out[:, c:2 * c].float(), ref) < 5e-3 and relerr(dv[id(out)][:, c:2 * c].float(), ref) < 5e-3
        assert float(dv[id(out)][:, :c].abs().max()) == 0 and float(dv[id(out)][:, 2 * c:].abs().max()) == 0


@pytest.mark.parametrize('rows,ld,rps,with_a,acc', [(37, 96, 0, 1, 0), (64, 768, 16, 1, 0), (50, 24, 10, 0, 1),
                                                    (200704, 96, 3136, 1, 0)])
def test_scale_rows_add(libs, rows, ld, rps, with_a, acc):
    a, b = rnd(rows, ld).to(BF16), rnd(rows, ld, seed=1).to(BF16)
    out = rnd(rows, ld, seed=2).to(BF16)
    rs = (torch.rand((rows + rps - 1) // rps, generator=torch.Generator().manual_seed(0)) < 0.7).float() / 0.7 if rps else None
    want = b.float() * (rs.repeat_interleave(rps)[:rows, None] if rps else 1.0) + (a.float() if with_a else 0) + \
        (out.float() if acc else 0)
    dv = both(libs, 'tok_scale_rows_add', lambda d: [d(a) if with_a else None, d(b), d(rs) if rps else None, rps, d(out), acc,
                                                     rows, ld, None])
    assert torch.equal(out, want.to(BF16))
    assert relerr(dv[id(out)].float(), want) < 3e-3          # fp32 fma contraction may move one bf16 ulp


@pytest.mark.parametrize('rows,c,k', [(1000, 96, 384), (802, 384, 96), (4096, 768, 3072), (77, 24, 40)])
@pytest.mark.parametrize('kind', [0, 1])
def test_gemm_activation_epilogues_equal_the_unfused_pair(libs, rows, c, k, kind):
    """tok_conv_fwd_act == tok_conv_fwd + tok_act_fwd and tok_conv_dgrad_act == tok_conv_dgrad + tok_act_bwd, bit for bit
    (the epilogue applies the activation to the bf16-rounded GEMM result, exactly what the separate launch reads)."""
    lib = libs[0]
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t_: t_.data_ptr()   # noqa: E731
    d = _desc(rows, 1, 1, c, k, 1, 1, 0)
    x = rnd(rows, c).to(BF16).cuda()
    w = (rnd(k, c, seed=1) * c ** -0.5).to(BF16).cuda()
    bias = rnd(k, seed=2).cuda()
    y0, y1, a0, a1 = (torch.empty(rows, k, dtype=BF16, device='cuda') for _ in range(4))
    assert lib.tok_conv_fwd(d, P(x), P(w), P(bias), P(y0), None, st) == 0
    assert lib.tok_act_fwd(kind, P(y0), P(a0), y0.numel(), st) == 0
    assert lib.tok_conv_fwd_act(d, P(x), P(w), P(bias), P(y1), P(a1), kind, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(a0, a1)
    # backward of the NEXT layer (k -> c2): its dgrad produces d(act(y)); fused: d(y) directly
    c2 = 64
    d2 = _desc(rows, 1, 1, k, c2, 1, 1, 0)
    g = rnd(rows, c2, seed=3).to(BF16).cuda()
    wd = (rnd(k, c2, seed=4) * c2 ** -0.5).to(BF16).cuda()           # dgrad pack [C=k][1][1][K=c2]
    da, dy0, dy1 = (torch.empty(rows, k, dtype=BF16, device='cuda') for _ in range(3))
    assert lib.tok_conv_dgrad(d2, P(g), P(wd), P(da), 0, st) == 0
    assert lib.tok_act_bwd(kind, P(da), P(y0), P(dy0), 0, da.numel(), st) == 0
    assert lib.tok_conv_dgrad_act(d2, P(g), P(wd), P(y0), kind, P(dy1), st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dy0, dy1)
    d3 = _desc(2, 8, 8, 16, 16, 3, 1, 1)
    assert lib.tok_conv_fwd_act(d3, P(x), P(w), None, P(y1), P(a1), kind, st) != 0         # 3x3: refused


@pytest.mark.parametrize('rows,c', [(1000, 96), (77, 192), (4133, 384), (256, 96), (70000, 96), (33000, 192), (16500, 384)])
@pytest.mark.parametrize('save', [False, True])
def test_fused_mlp_equals_the_two_gemm_launches(libs, rows, c, save):
    """tok_mlp_fwd == tok_conv_fwd_act (fc1 + GELU) + tok_conv_fwd (fc2), bit for bit, ragged row counts included (tiles of
    256 / 128 tokens, several tiles per workgroup at the larger counts); with pre / act given those rows are the unfused
    launches' too.  Against the fp32 restatement (fake_backend: torch matmul + F.gelu on the same rounding points) <= 1e-2."""
    lib, fake = libs
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t_: t_.data_ptr() if t_ is not None else None   # noqa: E731
    h = 4 * c
    assert lib.tok_mlp_serves(rows, c, h) == 1
    assert lib.tok_mlp_serves(rows, c, 2 * c) == 0 and lib.tok_mlp_serves(rows, 768, 3072) == 0 and lib.tok_mlp_serves(rows, 100, 400) == 0
    x = rnd(rows, c).to(BF16).cuda()
    w1 = (rnd(h, c, seed=1) * c ** -0.5).to(BF16).cuda()
    w2 = (rnd(c, h, seed=2) * h ** -0.5).to(BF16).cuda()
    b1, b2 = (rnd(h, seed=3) * 0.3).cuda(), (rnd(c, seed=4) * 0.3).cuda()
    d1, d2 = _desc(rows, 1, 1, c, h, 1, 1, 0), _desc(rows, 1, 1, h, c, 1, 1, 0)
    pre0, act0, pre1, act1 = (torch.full((rows, h), 7.0, dtype=BF16, device='cuda') for _ in range(4))
    y0, y1 = (torch.empty(rows, c, dtype=BF16, device='cuda') for _ in range(2))
    guard = torch.full((4096,), 3.0, dtype=BF16, device='cuda')          # lands right behind y1 in most allocators; checked below
    assert lib.tok_conv_fwd_act(d1, P(x), P(w1), P(b1), P(pre0), P(act0), 1, st) == 0, lib.tok_last_error()
    assert lib.tok_conv_fwd(d2, P(act0), P(w2), P(b2), P(y0), None, st) == 0, lib.tok_last_error()
    assert lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y1), P(pre1) if save else None, P(act1) if save else None,
                           rows, c, h, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    if save:
        assert torch.equal(pre0, pre1) and torch.equal(act0, act1)
    else:
        assert bool((pre1 == 7.0).all()) and bool((act1 == 7.0).all())
    assert bool((guard == 3.0).all())
    n = min(rows, 2048)
    yh = torch.empty(n, c, dtype=BF16)
    xc, w1c, w2c, b1c, b2c = x[:n].cpu(), w1.cpu(), w2.cpu(), b1.cpu(), b2.cpu()
    assert fake.tok_mlp_fwd(P(xc), P(w1c), P(b1c), P(w2c), P(b2c), P(yh), None, None, n, c, h, None) == 0
    assert relerr(y1[:n].float(), yh.float()) < 1e-2
    # pre without act (the recompute plan of training): the pre-activation rows only, nothing else is touched
    pre2 = torch.full((rows, h), 7.0, dtype=BF16, device='cuda')
    act1.fill_(7.0)
    assert lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y1), P(pre2), None, rows, c, h, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(pre0, pre2) and bool((act1 == 7.0).all()) and bool((guard == 3.0).all())
    # act without pre: refused
    assert lib.tok_mlp_fwd(P(x), P(w1), P(b1), P(w2), P(b2), P(y1), None, P(act1), rows, c, h, st) != 0


@pytest.mark.parametrize('rows,c', [(1000, 96), (77, 192), (4133, 384), (70000, 96), (33000, 192), (16500, 384)])
@pytest.mark.parametrize('acc', [0, 1])
def test_fused_mlp_backward_equals_the_two_dgrad_launches(libs, rows, c, acc):
    """tok_mlp_bwd_dx == tok_conv_dgrad_act (fc2, GELU') + tok_conv_dgrad (fc1), bit for bit: dx (fresh or accumulated onto
    an earlier contribution) and, when asked for, the d(pre) rows; against the fp32 restatement <= 1e-2."""
    lib, fake = libs
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t_: t_.data_ptr() if t_ is not None else None   # noqa: E731
    h = 4 * c
    dy = rnd(rows, c).to(BF16).cuda()
    pre = (rnd(rows, h, seed=5) * 1.5).to(BF16).cuda()
    w2d = (rnd(h, c, seed=1) * h ** -0.5).to(BF16).cuda()           # fc2 dgrad pack [hidden][c]
    w1d = (rnd(c, h, seed=2) * c ** -0.5).to(BF16).cuda()           # fc1 dgrad pack [c][hidden]
    base = rnd(rows, c, seed=6).to(BF16).cuda()
    d1, d2 = _desc(rows, 1, 1, c, h, 1, 1, 0), _desc(rows, 1, 1, h, c, 1, 1, 0)
    dpre0 = torch.empty(rows, h, dtype=BF16, device='cuda')
    dpre1 = torch.full((rows, h), 7.0, dtype=BF16, device='cuda')
    dx0, dx1, dx2 = base.clone(), base.clone(), base.clone()
    assert lib.tok_conv_dgrad_act(d2, P(dy), P(w2d), P(pre), 1, P(dpre0), st) == 0, lib.tok_last_error()
    assert lib.tok_conv_dgrad(d1, P(dpre0), P(w1d), P(dx0), acc, st) == 0, lib.tok_last_error()
    assert lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx1), acc, P(dpre1), rows, c, h, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1) and torch.equal(dpre0, dpre1)
    dpre1.fill_(7.0)
    assert lib.tok_mlp_bwd_dx(P(dy), P(w2d), P(pre), P(w1d), P(dx2), acc, None, rows, c, h, st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx2) and bool((dpre1 == 7.0).all())
    n = min(rows, 2048)
    dxh = base[:n].cpu().clone()
    dyc, prec, w2c, w1c = dy[:n].cpu(), pre[:n].cpu(), w2d.cpu(), w1d.cpu()
    assert fake.tok_mlp_bwd_dx(P(dyc), P(w2c), P(prec), P(w1c), P(dxh), acc, None, n, c, h, None) == 0
    assert relerr(dx1[:n].float(), dxh.float()) < 1e-2


@pytest.mark.parametrize('n,hs,ws,classes,hd,wd', [(2, 32, 64, 19, 128, 256), (1, 16, 24, 3, 64, 96), (2, 9, 13, 21, 36, 52),
                                                   (1, 8, 8, 8, 20, 28), (3, 128, 256, 19, 512, 1024)])
def test_upsample_ce_equals_interpolate_then_cross_entropy(libs, n, hs, ws, classes, hd, wd):
    """tok_upsample_ce_fwd / _bwd == tok_bilinear_fwd -> tok_softmax_ce_fwd and tok_softmax_ce_bwd -> tok_bilinear_bwd (the
    launches it replaces, whose full-resolution logits and gradient it never writes): loss, per-pixel lse, d(low) — scale
    factors 4, 2.5 and the HRNet-W48 head size; ignored pixels, a fresh and an accumulated gradient.  The interpolated value
    and d(upsampled logits) are rounded to bf16 exactly where the unfused launches store them, so the results agree to
    the contraction of one fp32 expression (measured: identical loss, d(low) <= 1e-3)."""
    lib = libs[0]
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t_: t_.data_ptr() if t_ is not None else None   # noqa: E731
    ld = (classes + 7) // 8 * 8
    assert lib.tok_upsample_ce_serves(classes, ld) == 1 and lib.tok_upsample_ce_serves(40, 40) == 0
    low = torch.zeros(n, hs, ws, ld, dtype=BF16, device='cuda')
    low[..., :classes] = (rnd(n, hs, ws, classes, seed=1) * 2).to(BF16).cuda()
    tgt = torch.randint(0, classes, (n, hd, wd), generator=torch.Generator().manual_seed(2)).cuda()
    tgt[:, :2] = 255
    tgt[:, :, -1] = 255
    rows = n * hd * wd
    gs = torch.tensor([0.7], device='cuda')
    # unfused
    up = torch.zeros(n, hd, wd, ld, dtype=BF16, device='cuda')
    assert lib.tok_bilinear_fwd(P(low), n, hs, ws, ld, ld, P(up), hd, wd, ld, 0, st) == 0, lib.tok_last_error()
    lse0, rl0 = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    loss0 = torch.zeros(_C.TOK_CE_LOSS_FLOATS, device='cuda')
    assert lib.tok_softmax_ce_fwd(P(up), P(tgt), rows, classes, ld, 255, P(lse0), P(rl0), P(loss0), st) == 0
    dup = torch.empty_like(up)
    assert lib.tok_softmax_ce_bwd(P(up), P(tgt), P(lse0), P(loss0), P(gs), rows, classes, ld, 255, P(dup), st) == 0
    base = torch.zeros(n, hs, ws, ld, dtype=BF16, device='cuda')
    base[..., :classes] = (rnd(n, hs, ws, classes, seed=3) * 1e-3).to(BF16).cuda()
    d0, d0a = torch.zeros_like(low), base.clone()
    assert lib.tok_bilinear_bwd(P(dup), n, hd, wd, ld, 0, P(d0), hs, ws, ld, ld, 0, st) == 0
    assert lib.tok_bilinear_bwd(P(dup), n, hd, wd, ld, 0, P(d0a), hs, ws, ld, ld, 1, st) == 0
    # fused
    lse1, rl1 = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
    loss1 = torch.zeros(_C.TOK_CE_LOSS_FLOATS, device='cuda')
    assert lib.tok_upsample_ce_fwd(P(low), n, hs, ws, classes, ld, hd, wd, P(tgt), 255, P(lse1), P(rl1), P(loss1), st) == 0, \
        lib.tok_last_error()
    d1, d1a = torch.full_like(low, 9.0), base.clone()
    assert lib.tok_upsample_ce_bwd(P(low), n, hs, ws, classes, ld, hd, wd, P(tgt), 255, P(lse1), P(loss1), P(gs), P(d1), 0, st) == 0, \
        lib.tok_last_error()
    assert lib.tok_upsample_ce_bwd(P(low), n, hs, ws, classes, ld, hd, wd, P(tgt), 255, P(lse1), P(loss1), P(gs), P(d1a), 1, st) == 0
    torch.cuda.synchronize()
    assert float(loss1[1]) == float(loss0[1]) == float((tgt != 255).sum())
    assert abs(float(loss1[0]) - float(loss0[0])) <= 1e-6 * abs(float(loss0[0]))
    assert float((lse1 - lse0).abs().max()) <= 1e-5 * float(lse0.abs().max())
    assert relerr(d1[..., :classes].float(), d0[..., :classes].float()) < 1e-3
    assert relerr(d1a[..., :classes].float(), d0a[..., :classes].float()) < 1e-3
    assert bool((d1[..., classes:] == 0).all())                       # padding channels of a fresh gradient are zeroed
    # against torch in fp32 on the same bf16 low-resolution logits
    lowf = low[..., :classes].float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(torch.nn.functional.interpolate(lowf, size=(hd, wd), mode='bilinear', align_corners=False),
                                            tgt, ignore_index=255)
    (ref * 0.7).backward()
    assert abs(float(loss1[0]) - float(ref)) < 3e-3 * abs(float(ref))
    assert relerr(d1[..., :classes].float().permute(0, 3, 1, 2), lowf.grad) < 1e-2


@pytest.mark.parametrize('n,h,w,c', [(2, 16, 16, 64), (1, 15, 17, 8), (3, 7, 9, 128), (4, 112, 112, 64)])
def test_fused_stem_pool_equals_the_unfused_chain(libs, n, h, w, c):
    """tok_bn_relu_maxpool_fwd == tok_bn_act_fwd + tok_maxpool3x3s2_fwd; tok_bn_pool_bwd_reduce / _apply ==
    tok_maxpool3x3s2_bwd + tok_bn_bwd_reduce / tok_bn_bwd_apply — bit for bit, partial rows included."""
    lib = libs[0]
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t_: None if t_ is None else t_.data_ptr()   # noqa: E731
    m = n * h * w
    p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = rnd(m, c).to(BF16).cuda()
    scale, shift = (rnd(c, seed=1) * 0.5 + 1).cuda(), (rnd(c, seed=2) * 0.3).cuda()
    mean, rstd = (rnd(c, seed=3) * 0.1).cuda(), (rnd(c, seed=4).abs() + 0.5).cuda()
    z = torch.empty(m, c, dtype=BF16, device='cuda')
    mask = torch.empty(m, c // 8, dtype=torch.uint8, device='cuda')
    pooled0, pooled1 = (torch.empty(n, p, q, c, dtype=BF16, device='cuda') for _ in range(2))
    idx0, idx1 = (torch.empty(n, p, q, c, dtype=torch.uint8, device='cuda') for _ in range(2))
    assert lib.tok_bn_act_fwd(P(y), P(scale), P(shift), None, 1, P(z), P(mask), m, c, st) == 0
    assert lib.tok_maxpool3x3s2_fwd(P(z), P(pooled0), P(idx0), n, h, w, c, st) == 0
    ypool = torch.empty(n, p, q, c, dtype=BF16, device='cuda')
    assert lib.tok_bn_relu_maxpool_fwd(P(y), P(scale), P(shift), n, h, w, c, P(pooled1), P(idx1), P(ypool), st) == 0, \
        lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(pooled0, pooled1) and torch.equal(idx0, idx1)
    g = rnd(n * p * q, c, seed=5).to(BF16).cuda()
    dz = torch.empty(m, c, dtype=BF16, device='cuda')
    rows = lib.tok_bn_bwd_rows(m, c)
    part0, part1 = (torch.zeros(2, rows, c, device='cuda') for _ in range(2))
    assert lib.tok_maxpool3x3s2_bwd(P(g), P(idx0), P(dz), 0, n, h, w, c, st) == 0
    assert lib.tok_bn_bwd_reduce(P(dz), P(y), P(mask), P(scale), P(shift), P(mean), P(rstd), 1, m, c, P(part0), st) == 0
    assert lib.tok_bn_pool_bwd_reduce(P(g), P(idx0), P(y), P(scale), P(shift), P(mean), P(rstd), n, h, w, c, P(part1),
                                      st) == 0, lib.tok_last_error()
    coef = torch.stack([rnd(c, seed=6), rnd(c, seed=7) * 0.1, rnd(c, seed=8) * 0.01]).cuda()
    dy0, dy1 = (torch.empty(m, c, dtype=BF16, device='cuda') for _ in range(2))
    assert lib.tok_bn_bwd_apply(P(dz), P(y), P(mask), P(scale), P(shift), P(coef), 1, P(dy0), None, 0, m, c, st) == 0
    assert lib.tok_bn_pool_bwd_apply(P(g), P(idx0), P(y), P(scale), P(shift), P(coef), n, h, w, c, P(dy1), st) == 0, \
        lib.tok_last_error()
    # pooled-domain sums: equal up to the bf16 rounding of positions that several windows hit
    rows_p = lib.tok_bn_bwd_rows(n * p * q, c)
    part2 = torch.zeros(2, rows_p, c, device='cuda')
    assert lib.tok_bn_pool_bwd_reduce_pooled(P(g), P(pooled1), P(ypool), P(mean), P(rstd), n * p * q, c, P(part2), st) == 0, \
        lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(part0, part1) and torch.equal(dy0, dy1)
    ih = 2 * torch.arange(p).view(1, p, 1, 1) - 1 + idx1.cpu().long() // 3
    iw = 2 * torch.arange(q).view(1, 1, q, 1) - 1 + idx1.cpu().long() % 3
    want_y = torch.gather(y.cpu().view(n, h * w, c), 1, (ih * w + iw).view(n, p * q, c)).view(n, p, q, c)
    assert torch.equal(ypool.cpu(), want_y)
    # the pooled-domain sums are the exact ones (no bf16 rounding of a position's summed gradient in between); the
    # position-domain sums of the unfused chain scatter around them by that rounding: ~2^-9 |g| sqrt(positions)
    gz = g.double().cpu().view(-1, c) * (pooled1.double().cpu().view(-1, c) > 0)
    xh = (ypool.double().cpu().view(-1, c) - mean.double().cpu()) * rstd.double().cpu()
    exact = torch.stack([gz.sum(0), (gz * xh).sum(0)])
    b = part2.double().sum(1).cpu()
    assert float((b - exact).abs().max()) < 1e-4 * float(exact.abs().max()) + 1e-4
    a = part0.double().sum(1).cpu()
    assert float((a - exact).abs().max()) < 2.0 ** -7 * (m ** 0.5) * 4


# ---- OCR head: pixel <-> class products ---------------------------------------------------------------------------------------
@pytest.mark.parametrize('images,n,k,c,pad', [(2, 100, 7, 48, 0), (3, 384, 19, 128, 0), (1, 1000, 19, 64, 8), (2, 33, 3, 20, 8), (2, 700, 19, 512, 0)])
def test_ocr_building_blocks(libs, images, n, k, c, pad):
    cp = (c + 7) // 8 * 8 + pad
    x = rnd(images * n, cp).to(BF16)
    m = rnd(images * k, cp, seed=1).to(BF16)
    w = rnd(images, n, k, seed=2).softmax(-1).contiguous()
    out = torch.empty(images, n, k)
    dv = both(libs, 'tok_pix_class_matmul', lambda d: [d(x), cp, d(m), cp, images, n, k, c, 0.5, d(out), None])
    ref = 0.5 * x.float().view(images, n, cp)[..., :c] @ m.float().view(images, k, cp)[..., :c].transpose(1, 2)
    assert relerr(out, ref) < 1e-5 and relerr(dv[id(out)], ref) < 1e-4
    for acc in (0, 1):
        o = rnd(images * n, cp, seed=3).to(BF16)
        o0 = o.clone()
        dv = both(libs, 'tok_class_pix_expand', lambda d: [d(w), d(m), cp, images, n, k, c, 2.0, d(o), cp, acc, None])
        ref = (2.0 * w @ m.float().view(images, k, cp)[..., :c]).reshape(images * n, c) + (o0.float()[:, :c] if acc else 0)
        assert relerr(o.float()[:, :c], ref) < 5e-3 and relerr(dv[id(o)].float()[:, :c], ref) < 5e-3
    lib = libs[0]
    chunks = lib.tok_weighted_pool_chunks(n)
    part = torch.empty(images, chunks, k, c)
    for acc in (0, 1):
        o = rnd(images * k, cp, seed=4).to(BF16)
        o0 = o.clone()
        dv = both(libs, 'tok_weighted_pool', lambda d: [d(w), d(x), cp, images, n, k, c, 0.25, d(part), d(o), cp, acc, None])
        ref = (0.25 * w.transpose(1, 2) @ x.float().view(images, n, cp)[..., :c]).reshape(images * k, c) + \
            (o0.float()[:, :c] if acc else 0)
        assert relerr(o.float()[:, :c], ref) < 5e-3 and relerr(dv[id(o)].float()[:, :c], ref) < 5e-3
    logits = (rnd(images * n, cp, seed=5) * 2).to(BF16)
    p = torch.empty(images, n, k)
    dv = both(libs, 'tok_softmax_cols_fwd', lambda d: [d(logits), cp, images, n, k, 1.0, d(p), None])
    ref = logits.float().view(images, n, cp)[..., :k].softmax(1)
    assert relerr(p, ref) < 1e-5 and relerr(dv[id(p)], ref) < 1e-4
    dp = rnd(images, n, k, seed=6)
    dl = torch.ones(images * n, cp, dtype=BF16)
    dv = both(libs, 'tok_softmax_cols_bwd', lambda d: [d(p), d(dp), images, n, k, 1.0, d(dl), cp, 0, None])
    assert relerr(dv[id(dl)].float(), dl.float()) < 5e-3 and float(dv[id(dl)][:, k:].abs().max()) == 0
    sm = torch.empty(images, n, k)
    dv = both(libs, 'tok_softmax_rows_f32', lambda d: [d(out), images * n, k, d(sm), None])
    assert relerr(dv[id(sm)], out.softmax(-1)) < 1e-5
    dx = torch.empty(images, n, k)
    dv = both(libs, 'tok_softmax_rows_bwd_f32', lambda d: [d(sm), d(dp), images * n, k, d(dx), None])
    assert relerr(dv[id(dx)], dx) < 1e-5
    s = rnd(images, c, seed=7)
    y = torch.empty(images * n, cp, dtype=BF16)
    dv = both(libs, 'tok_channel_scale', lambda d: [d(x), d(s), d(y), 0, images, n, c, cp, None])
    assert relerr(dv[id(y)].float(), y.float()) < 5e-3


@pytest.mark.parametrize('n,h,w,c,ld', [(2, 8, 8, 96, 96), (1, 7, 9, 20, 24), (3, 14, 14, 384, 384)])
def test_depthwise_3x3(libs, n, h, w, c, ld):
    """ConvPosEnc.proj (davit.py:101-106): forward, data gradient (flipped taps), weight / bias gradient."""
    x = rnd(n * h * w, ld).to(BF16)
    x[:, c:] = 0
    wt, b = rnd(c, 9, seed=1) * 0.3, rnd(c, seed=2)
    out = torch.empty(n * h * w, ld, dtype=BF16)
    dv = both(libs, 'tok_dwconv3x3', lambda d: [d(x), d(wt), d(b), d(out), 0, 0, n, h, w, c, ld, None])
    ref = torch.nn.functional.conv2d(x.float().view(n, h, w, ld)[..., :c].permute(0, 3, 1, 2), wt.view(c, 1, 3, 3), b, padding=1,
                                     groups=c).permute(0, 2, 3, 1).reshape(-1, c)
    assert relerr(out.float()[:, :c], ref) < 5e-3 and relerr(dv[id(out)].float()[:, :c], ref) < 5e-3
    g = rnd(n * h * w, ld, seed=3).to(BF16)
    g[:, c:] = 0
    dx = torch.empty(n * h * w, ld, dtype=BF16)
    dv = both(libs, 'tok_dwconv3x3', lambda d: [d(g), d(wt), None, d(dx), 0, 1, n, h, w, c, ld, None])
    assert relerr(dv[id(dx)].float(), dx.float()) < 5e-3
    lib = libs[0]
    part = torch.empty(lib.tok_dwconv3x3_wgrad_blocks(n, h), c, 10)
    dw, db = torch.empty(c, 9), torch.empty(c)
    dv = both(libs, 'tok_dwconv3x3_wgrad', lambda d: [d(x), d(g), n, h, w, c, ld, d(part), d(dw), d(db), 0, None])
    assert relerr(dv[id(dw)], dw) < 1e-4 and relerr(dv[id(db)], db) < 1e-4


@pytest.mark.parametrize('p,k', [(64, 256), (128, 512), (256, 512), (72, 40), (320, 640)])
def test_bn_gram_finalize(libs, p, k):
    """BatchNorm statistics of y = z W^T from the Gram matrix of z (the fused residual unit): mean / rstd / scale / shift /
    running statistics and the kept product W Z, for the in-kernel product (p <= 256) and the tiled GEMM path."""
    m = 5000
    z = rnd(m, p).to(BF16).float() + 0.3
    Z = (z.double().t() @ z.double()).float()
    zsum = z.double().sum(0).float()
    w = rnd(k, p, scale=p ** -0.5, seed=2)
    gamma, beta = rnd(k, seed=3) + 1, rnd(k, seed=4)
    rm, rv = rnd(k, seed=5), rnd(k, seed=6).abs() + 0.5
    nbt = torch.zeros(1, dtype=torch.int64)
    mean, rstd, scale, shift = (torch.zeros(k) for _ in range(4))
    wz = torch.zeros(k, p)
    dv = both(libs, 'tok_bn_gram_finalize',
              lambda f: [f(Z), f(zsum), f(w), m, p, k, f(gamma), f(beta), f(rm), f(rv), f(nbt), 0.1, 1e-5, f(mean), f(rstd),
                         f(scale), f(shift), f(wz), None])
    for t in (mean, rstd, scale, shift, rm, rv):
        assert relerr(dv[id(t)], t) < 1e-4
    assert relerr(dv[id(wz)], wz) < 1e-5
    assert int(dv[id(nbt)].item()) == 1 and int(nbt.item()) == 1


@pytest.mark.parametrize('p,k', [(64, 256), (256, 512)])
def test_bn_gram_finalize_large_mean_channel(libs, p, k):
    """an input channel with |mean| >> std (here mean 2, std 0.05 — bf16 storage leaves ~0.05 of spread —, 200 000 rows, Gram matrix in fp32) must not
    lose the variance to E[y^2] - mean^2 cancellation: the kernel centres the Gram row in fp64 before the w products."""
    m = 200000
    g = torch.Generator().manual_seed(p + k)
    z = (torch.randn(m, p, generator=g) * 0.05 + 2.0).to(BF16).float()
    Z = (z.t() @ z)                                             # fp32 accumulation, as the weight-gradient launch leaves it
    zsum = z.double().sum(0).float()
    w = rnd(k, p, scale=p ** -0.5, seed=2)
    gamma, beta = torch.ones(k), torch.zeros(k)
    rm, rv = torch.zeros(k), torch.ones(k)
    nbt = torch.zeros(1, dtype=torch.int64)
    mean, rstd, scale, shift = (torch.zeros(k, device=DEV) for _ in range(4))
    wz = torch.zeros(k, p, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    d = lambda t: t.to(DEV)  # noqa: E731
    Zd, zd, wd, gd, bd, rmd, rvd, nd = d(Z), d(zsum), d(w), d(gamma), d(beta), d(rm), d(rv), d(nbt)
    assert libs[0].tok_bn_gram_finalize(Zd.data_ptr(), zd.data_ptr(), wd.data_ptr(), m, p, k, gd.data_ptr(), bd.data_ptr(),
                                        rmd.data_ptr(), rvd.data_ptr(), nd.data_ptr(), 0.1, 1e-5, mean.data_ptr(),
                                        rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), wz.data_ptr(), st) == 0
    torch.cuda.synchronize()
    y = z.double() @ w.to(BF16).double().t()                    # what the unit's GEMM produces, exactly
    var = y.var(0, unbiased=False)
    want = 1.0 / torch.sqrt(var + 1e-5)
    # fp32 Gram entries carry ~1e-7 relative noise on 1600 * m; after centring, the variance (~0.05^2 |w|^2) survives
    assert relerr(rstd.cpu().double(), want) < 5e-2
    assert float(rstd.max()) < 0.5 / (1e-5 ** 0.5)             # nowhere near the clamp value 1 / sqrt(eps)


@pytest.mark.parametrize('decoupled', [0, 1])
def test_adam_capturable(libs, decoupled):
    """Device-side step count: three successive launches + tok_step_advance equal the host-step kernel (and torch's update
    restated in tests/fake_backend.py) step for step."""
    lib, fake = libs
    n = 5000
    p0, m0, v0 = rnd(n), torch.zeros(n), torch.zeros(n)
    st = torch.cuda.current_stream().cuda_stream
    pd, md, vd = p0.to(DEV), m0.to(DEV), v0.to(DEV)
    ph, mh, vh = p0.clone(), m0.clone(), v0.clone()
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for it in range(3):
        g = rnd(n, seed=10 + it)
        gd = g.to(DEV)
        assert lib.tok_adam_step_capturable(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), None, n, 1e-2, 0.9, 0.999,
                                            1e-8, 0.05, decoupled, step.data_ptr(), 0, st) == 0, lib.tok_last_error()
        assert lib.tok_step_advance(step.data_ptr(), st) == 0
        assert fake.tok_adam_step(ph.data_ptr(), g.data_ptr(), mh.data_ptr(), vh.data_ptr(), None, n, 1e-2, 0.9, 0.999, 1e-8,
                                  0.05, decoupled, it + 1, 0, None) == 0
    torch.cuda.synchronize()
    assert int(step.item()) == 3
    assert relerr(pd, ph) < 1e-6 and relerr(md, mh) < 1e-6 and relerr(vd, vh) < 5e-5     # (v: fma vs mul + addcmul rounding)


@pytest.mark.parametrize('m,c,relu', [(5000, 64, 1), (777, 256, 1), (130, 1024, 0), (64, 8, 1)])
def test_bn_act_fwd_colsum(libs, m, c, relu):
    """The activation pass that also leaves per-block column sums of its output: same `out` / mask as tok_bn_act_fwd, and
    the folded partials equal the column sums of the STORED bf16 values."""
    lib, _ = libs
    st = torch.cuda.current_stream().cuda_stream
    y = rnd(m, c).to(BF16).to(DEV)
    scale, shift = (rnd(c, seed=1) * 0.5 + 1).to(DEV), rnd(c, seed=2).to(DEV)
    out1, out2 = torch.empty_like(y), torch.empty_like(y)
    mk1 = torch.zeros(m, c // 8, dtype=torch.uint8, device=DEV)
    mk2 = torch.zeros_like(mk1)
    rows = lib.tok_bn_act_fwd_colsum_rows(m, c)
    assert rows > 0
    part = torch.full((rows, c), float('nan'), device=DEV)
    assert lib.tok_bn_act_fwd(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, relu, out1.data_ptr(), mk1.data_ptr(), m, c,
                              st) == 0
    assert lib.tok_bn_act_fwd_colsum(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None, relu, out2.data_ptr(), mk2.data_ptr(),
                                     m, c, part.data_ptr(), st) == 0, lib.tok_last_error()
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(mk1, mk2)
    assert not torch.isnan(part).any()
    assert relerr(part.sum(0), out2.float().sum(0)) < 1e-5


@pytest.mark.parametrize('p,k', [(64, 256), (128, 512), (256, 512), (72, 136)])
def test_bn3_bwd_prepare(libs, p, k):
    """Backward coefficients of the fused residual unit from G = dz^T z, W, W Z and the column sums: dgamma / dbeta, the three
    apply coefficients, dW, and the operands of d(input) = dz Wa + z Wb + c — block-per-row product (p in {64, 128, 256}) and the
    tiled product + split reduce."""
    lib, fake = libs
    m = 4000
    z = rnd(m, p).to(BF16).float() + 0.2
    dz = rnd(m, k, seed=1).to(BF16).float()
    w = rnd(k, p, scale=p ** -0.5, seed=2)
    wq = w.to(BF16).float()
    G = (dz.double().t() @ z.double()).float()
    Z = (z.double().t() @ z.double()).float()
    wz = (wq.double() @ Z.double()).float()
    zsum = z.double().sum(0).float()
    rows = 3
    partial = torch.zeros(2, rows, k)
    partial[0, 1] = dz.double().sum(0).float()
    y = z @ wq.t()
    mean = y.mean(0)
    rstd = 1.0 / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
    gamma = rnd(k, seed=3) + 1
    outs = dict(dgamma=torch.zeros(k), dbeta=torch.zeros(k), coef=torch.zeros(3, k), dw=torch.zeros(k, p),
                wa=torch.zeros(p, k, dtype=BF16), wb=torch.zeros(p, p, dtype=BF16), cvec=torch.zeros(p))
    ws = torch.zeros(int(lib.tok_bn3_bwd_prepare_ws_floats(p, k)) + 16)
    dv = both(libs, 'tok_bn3_bwd_prepare',
              lambda f: [f(G), f(w), f(wz), f(zsum), f(partial), rows, m, p, k, f(gamma), f(mean), f(rstd), f(outs['dgamma']),
                         f(outs['dbeta']), 0, f(outs['coef']), f(outs['dw']), 0, f(outs['wa']), f(outs['wb']), f(outs['cvec']),
                         f(ws), None])
    for name in ('dgamma', 'dbeta', 'coef', 'dw', 'cvec'):
        assert relerr(dv[id(outs[name])], outs[name]) < 2e-4, name
    for name in ('wa', 'wb'):
        assert relerr(dv[id(outs[name])].float(), outs[name].float()) < 4e-3, name


def test_gemm256_tile_kernel_is_bit_identical_to_the_default_kernels():
    """csrc/gemm256.hip (256 x 256 tiles, eight waves; off by default, TOK_GEMM256=1) against the kernels that serve the same
    pointwise layers by default: forward + batch statistics, data gradient fresh / accumulated / with BatchNorm-backward sums / with the ReLU-masked
    store on ResNet-50 layer-3/4 and SwinV2-T stage-3/4 shapes.  The knob is read once per process, so the two arms run as subprocesses
    (tools/ubench/g256_check.py) and the comparison is parsed from its report: outputs bit-identical, statistics to fp32 summation order."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ubench', 'g256_check.py')], capture_output=True, text=True,
                         timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('M,K,N=')]
    assert len(lines) >= 10, out.stdout[-2000:]
    rows_new = re.findall(r'\[new\].*rows (\d+)/(\d+)', out.stdout)
    rows_old = re.findall(r'\[old\].*rows (\d+)/(\d+)', out.stdout)
    assert rows_new and rows_new != rows_old          # the two arms really ran different kernels (other statistics rows)
    for l in lines:
        v = dict(re.findall(r'\b(yact|ya|y|s|dxa|dxs|dxm|dxg|dx|pm|p) ([0-9.e+-]+)', l.split('new vs old')[1].split('y-vs-fp32')[0]))
        assert all(float(v[n]) == 0.0 for n in ('y', 'dx', 'dxa', 'dxs', 'dxm', 'ya', 'yact', 'dxg')), l
        assert float(v['s']) < 1e-5 and float(v['p']) < 1e-5 and float(v['pm']) < 1e-5, l


def test_wgrad_256_tile_kernel_matches_the_128_tile_plan():
    """conv_wgrad_ring8_kernel<256, 256> (eight waves; serves pointwise layers over >= 200 k pixels by default, every eligible layer
    with TOK_WGRAD_256=2) against the 128 x 128 ring plan: dW, accumulated dW and the bias column sums agree to fp32 summation
    order (different split-M partition), and with the fp32 product where the host can form it.  Subprocess arms: the knob is
    read once per process (tools/ubench/wgrad256_check.py)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'ubench', 'wgrad256_check.py')], capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('M,Ktot,k=')]
    assert len(lines) >= 12, out.stdout[-2000:]
    ws_new = re.findall(r'\[new\].*workspace\s+([0-9.]+) MB', out.stdout)
    ws_old = re.findall(r'\[old\].*workspace\s+([0-9.]+) MB', out.stdout)
    assert ws_new and ws_new != ws_old              # other split-M plans: the two arms ran different kernels
    for l in lines:
        v = dict(re.findall(r'(dw|db|dwa) ([0-9.e+-]+)', l.split('new vs old')[1].split('db-vs-fp32')[0]))
        assert float(v['dw']) < 5e-6 and float(v['db']) < 5e-6 and float(v['dwa']) < 5e-6, l
        m = re.search(r'dW-vs-fp32 new ([0-9.e+-]+)', l)
        if m:
            assert float(m.group(1)) < 1e-5, l
