"""The ring-less pointwise GEMM (csrc/pw_gemm.hip: pw_stream_kernel, round 5) against the ring kernel it replaces on the
write-heavy streaming layers: EVERY output bit for bit — forward (+ BatchNorm statistics rows), data gradient plain /
accumulating / with the BatchNorm-backward sums (with and without ReLU bits) / ReLU-masked store / bias / the absorbed
half-resolution gradient of a strided shortcut — on ragged pixel counts, reduction depths 64 and 128, 64 ... 512 outputs.
The kernel choice is a process-wide knob (TOK_PW_STREAM, read once), so the same script runs in two subprocesses
(ring: TOK_PW_STREAM=0, stream on every layer it can run: =2, both with the ring's row threshold lowered to 1) and the dumps
are compared.  Each entry point is separately checked against the fp32 restatement by tests/test_kernels_gpu.py."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(3, 18, 22, 256, 64), (2, 16, 16, 512, 128), (5, 10, 14, 64, 64), (2, 28, 28, 256, 128), (1, 56, 56, 128, 64),
         (32, 56, 56, 256, 64)]          # (n, h, w, c = channels of x / dx, k = channels of y / dy); the last one: 100 352 rows


def _worker(out_path: str):
    sys.path.insert(0, os.path.dirname(HERE))
    from torchok_amd import _C
    lib = _C.load_library()
    st = torch.cuda.current_stream().cuda_stream
    BF = torch.bfloat16
    P = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    res = {}
    for ci, (n, h, w, c, k) in enumerate(CASES):
        g = torch.Generator(device='cuda').manual_seed(100 + ci)
        m = n * h * w
        rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)  # noqa: E731
        # forward k -> c as a pointwise layer with a SHORT reduction (x has k channels): out has c channels
        df = _C.ConvDesc(n, h, w, k, c, 1, 1, h, w, 1, 0, 1)
        x = rnd(m, k).to(BF)
        wf = (rnd(c, k) * k ** -0.5).to(BF)
        y = torch.zeros(m, c, dtype=BF, device='cuda')
        rows = lib.tok_conv_fwd_stat_rows(ctypes.byref(df))
        stats = torch.zeros(2, rows, c, device='cuda')
        assert lib.tok_conv_fwd(ctypes.byref(df), P(x), P(wf), None, P(y), P(stats), st) == 0, lib.tok_last_error()
        res[f'{ci}.fwd.y'], res[f'{ci}.fwd.stats'] = y, stats
        # data gradient of the layer c -> k (1x1): dy has k channels, dx has c
        d = _C.ConvDesc(n, h, w, c, k, 1, 1, h, w, 1, 0, 1)
        dy = rnd(m, k).to(BF)
        wd = (rnd(c, k) * k ** -0.5).to(BF)          # dgrad pack [c][1][1][k]
        prev = rnd(m, c).to(BF)
        bn_y = rnd(m, c).to(BF)
        mask = torch.randint(0, 256, (m, c // 8), dtype=torch.uint8, device='cuda', generator=g)
        rows = lib.tok_conv_dgrad_stat_rows(ctypes.byref(d))
        for acc in (0, 1):
            dx = prev.clone()
            assert lib.tok_conv_dgrad(ctypes.byref(d), P(dy), P(wd), P(dx), acc, st) == 0, lib.tok_last_error()
            res[f'{ci}.dgrad{acc}'] = dx
        for wm in (0, 1):
            dx, part = prev.clone(), torch.zeros(2, rows, c, device='cuda')
            assert lib.tok_conv_dgrad_bnstats(ctypes.byref(d), P(dy), P(wd), P(dx), 1, P(bn_y), P(mask) if wm else None, P(part),
                                              st) == 0, lib.tok_last_error()
            res[f'{ci}.bnstats{wm}.dx'], res[f'{ci}.bnstats{wm}.part'] = dx, part
        dx, part = prev.clone(), torch.zeros(2, rows, c, device='cuda')
        assert lib.tok_conv_dgrad_maskstore(ctypes.byref(d), P(dy), P(wd), P(dx), 1, P(mask), P(part), st) == 0, lib.tok_last_error()
        res[f'{ci}.maskstore.dx'], res[f'{ci}.maskstore.part'] = dx, part
        bias = rnd(c)
        dx, part = prev.clone(), torch.zeros(2, rows, c, device='cuda')
        assert lib.tok_conv_dgrad_bias(ctypes.byref(d), P(dy), P(wd), P(bias), P(dx), 1, P(bn_y), P(mask), P(part), st) == 0, \
            lib.tok_last_error()
        res[f'{ci}.bias.dx'], res[f'{ci}.bias.part'] = dx, part
        if lib.tok_conv_dgrad_subacc_ok(ctypes.byref(d)):
            sub = rnd(n, (h + 1) // 2, (w + 1) // 2, c).to(BF)
            for ms in (0, 1):
                dx, part = torch.zeros(m, c, dtype=BF, device='cuda'), torch.zeros(2, rows, c, device='cuda')
                assert lib.tok_conv_dgrad_subacc(ctypes.byref(d), P(dy), P(wd), P(dx), P(sub), None if ms else P(bn_y), P(mask),
                                                 P(part), ms, st) == 0, lib.tok_last_error()
                res[f'{ci}.subacc{ms}.dx'], res[f'{ci}.subacc{ms}.part'] = dx, part
    torch.cuda.synchronize()
    torch.save({k_: v.cpu() for k_, v in res.items()}, out_path)
    print('PW-STREAM-DUMP-OK', len(res))


@pytest.mark.timeout(900)
def test_stream_kernel_is_bit_identical_to_the_ring(tmp_path):
    sys.path.insert(0, os.path.dirname(HERE))
    from torchok_amd import _C
    if not _C.load_library().tok_built_with_experiments():
        pytest.skip('pw_stream_kernel is compiled only with TOK_BUILD_EXPERIMENTS=1 (bit-identical to the ring kernel, slower in '
                    'every mode a training step uses: profiles/r05_pw_stream_probe.txt)')
    dumps = []
    for flag in ('0', '2'):
        out = str(tmp_path / f'dump{flag}.pt')
        env = dict(os.environ, TOK_PW_STREAM=flag, TOK_PW_RING_MIN_ROWS='1')
        p = subprocess.run([sys.executable, __file__, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                           timeout=400)
        assert p.returncode == 0 and 'PW-STREAM-DUMP-OK' in p.stdout, p.stdout[-3000:]
        dumps.append(torch.load(out))
    ring, stream = dumps
    assert ring.keys() == stream.keys() and len(ring) >= 60
    assert any('subacc' in k for k in ring)
    for k in ring:
        assert torch.equal(ring[k], stream[k]), (k, float((ring[k].float() - stream[k].float()).abs().max()))


if __name__ == '__main__':
    _worker(sys.argv[1])
