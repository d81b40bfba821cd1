"""Oracle parity at the DEFAULT dispatch thresholds.

tests/conftest.py lowers TOK_CONV_WIN_MIN_TILES / TOK_MLP_MIN_ROWS so that the small parity shapes reach the big-tile kernels;
the kernel-selection rules the benchmark runs under (UNIT3_MIN_ROWS = 100 k rows, the pointwise ring from 100 k rows,
conv_win from 128 tiles, gemm256_geometry: reduction >= 384 and >= 128 tiles, tok_mlp_serves from 32 768 rows) were therefore
only reached by the full-size property tests, which compare the path with itself.  Here mid-size units run in a SUBPROCESS
whose environment carries none of the overrides (the library reads its knobs once per process) and are compared with the
oracle (fp32 and bf16 autocast, tests/test_units_gpu.py's gate):

  * ResNet-50 layer1-style bottleneck 256 -> 64 -> 256 at 56 x 56, batch 32 (100 352 rows): fused residual unit, pointwise
    ring, conv_win<64, 64> — all at their default thresholds;
  * layer2-style stride-2 bottleneck with projection shortcut 256 -> 128 -> 512, 56 -> 28 px, batch 32;
  * layer3-style bottleneck 1024 -> 256 -> 1024 at 14 x 14, batch 176 (34 496 rows): gemm256 (reduction 1024, 135 tiles) and
    conv_win<16, 128> (135 tiles);
  * SwinV2 stage-1 block (C = 96, window 7, 56 x 56 tokens, batch 16 = 50 176 token rows): fused Mlp at its default row
    threshold, 128-wide token GEMMs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(case: str):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch
    import torch.nn as nn
    import oracle.timm_min as TM
    from test_units_gpu import _bf, _ours_map, _ref_unit, _round_weights_
    from test_units_real_gpu import _gate
    from torchok_amd import _C
    from torchok_amd.engine import functional as EF
    from torchok_amd.models.backbones import resnet as PR
    for k in ('TOK_CONV_WIN_MIN_TILES', 'TOK_MLP_MIN_ROWS', 'TOK_GEMM256', 'TOK_UNIT3_MIN_ROWS'):
        assert k not in os.environ, k
    assert EF.UNIT3_MIN_ROWS == 100000
    lib = _C.load_library()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(5)
    if case.startswith('bottleneck'):
        _, cin, planes, stride, hw, batch = case.split(':')
        cin, planes, stride, hw, batch = int(cin), int(planes), int(stride), int(hw), int(batch)
        ds = None
        if stride != 1 or cin != planes * 4:
            ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        blk = TM.Bottleneck(cin, planes, stride=stride, downsample=ds).train()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75)
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        _round_weights_(blk)
        x = _bf(torch.relu(torch.randn(batch, cin, hw, hw, generator=g)))        # a block input is a ReLU output
        ho = hw // stride
        gout = _bf(torch.randn(batch, planes * 4, ho, ho, generator=g) * 1e-3)
        # which kernels own the layers at this size, under the DEFAULT rules (the point of the test)
        d1 = _C.ConvDesc(batch, hw, hw, cin, planes, 1, 1, hw, hw, 1, 0, 1)
        rows = batch * hw * hw
        print(f'[default dispatch] {case}: rows {rows}, conv1 stat rows {lib.tok_conv_fwd_stat_rows(d1)}, '
              f'fused residual unit: {rows // (stride * stride) >= EF.UNIT3_MIN_ROWS}')
        r32, rac = _ref_unit(blk, x, gout), _ref_unit(blk, x, gout, autocast=True)
        ours_ds = None
        if ds is not None:
            import copy
            ours_ds = nn.Sequential(copy.deepcopy(ds[0]), copy.deepcopy(ds[1]))
        ours_blk = PR.Bottleneck(cin, planes, stride=stride, downsample=ours_ds)
        ours_blk.load_state_dict(blk.state_dict())
        ours_blk.cuda().train()
        ours = _ours_map(lambda r, ins: ours_blk(ins[0]), x, gout, 'cuda', ours_blk)
        _gate(f'default-dispatch {case}', ours, r32, rac)
    else:
        import oracle.swin_ref as S
        from test_units_gpu import _check, _ours_tokens
        from torchok_amd.models.backbones import swin as PS
        _, dim, heads, res, batch, shift = case.split(':')
        dim, heads, res, batch, shift = int(dim), int(heads), int(res), int(batch), int(shift)
        blk = S.SwinTransformerBlock(dim=dim, input_resolution=(res, res), num_heads=heads, window_size=7, shift_size=shift).train()
        _round_weights_(blk)
        x = _bf(torch.randn(batch, res * res, dim, generator=g))
        gout = _bf(torch.randn(batch, res * res, dim, generator=g) * 1e-3)
        assert lib.tok_mlp_serves(batch * res * res, dim, 4 * dim) == 1          # the fused Mlp at its default row threshold
        r32, rac = _ref_unit(blk, x, gout), _ref_unit(blk, x, gout, autocast=True)
        ours_blk = PS.SwinTransformerBlock(dim=dim, input_resolution=(res, res), num_heads=heads, window_size=7,
                                           shift_size=shift, drop_path=0.0)
        missing = ours_blk.load_state_dict(blk.state_dict(), strict=False)
        assert not missing.missing_keys, missing
        ours_blk.cuda().train()
        ours = _ours_tokens(lambda r, t, b: ours_blk.run(r, t, b), x, gout, 'cuda', ours_blk)
        # (composite gate of tests/test_units_gpu.py::test_swin_block_unit: the position-bias MLP holds a ReLU behind a bf16 Linear)
        _check(f'default-dispatch {case}', ours, r32, rac, composite=True)
    print('DEFAULT-DISPATCH-OK')


CASES = ['bottleneck:256:64:1:56:32', 'bottleneck:256:128:2:56:32', 'bottleneck:1024:256:1:14:176', 'swin:96:3:56:16:3']


@pytest.mark.parametrize('case', CASES)
@pytest.mark.timeout(900)
def test_unit_against_oracle_at_default_thresholds(case):
    env = {k: v for k, v in os.environ.items() if not k.startswith('TOK_') or k in ('TOK_LIB',)}
    p = subprocess.run([sys.executable, __file__, case], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=880)
    print(p.stdout[-6000:])
    assert p.returncode == 0 and 'DEFAULT-DISPATCH-OK' in p.stdout, p.stdout[-3000:]


if __name__ == '__main__':
    _worker(sys.argv[1])
