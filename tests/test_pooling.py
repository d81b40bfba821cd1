"""Row a6: Pooling / PoolingLinear for every pool type of the reference (`poolings/classification/pooling.py:7-12`,
`linear.py:8-25`; [timm] SelectAdaptivePool2d: avg / max / avgmax / catavgmax) against outputs and gradients of the
reference's OWN files (tests/golden/pooling_modes.npz, written by tests/golden/gen_golden.py --pooling-only), on the host
stand-in and, marked gpu, on libtok_gfx950.so.  The oracle restatement is pinned to the same vectors."""
import os

import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import rel_err
from oracle import timm_min

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pooling_modes.npz'))
TYPES = ('avg', 'max', 'avgmax', 'catavgmax')


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


@pytest.mark.parametrize('pt', TYPES)
def test_oracle_pool_equals_reference(pt):
    x = torch.from_numpy(G['x']).requires_grad_(True)
    y = timm_min.SelectAdaptivePool2d(1, pt, flatten=True)(x)
    assert torch.equal(y.detach(), torch.from_numpy(G[f'{pt}_y']))
    (y * torch.from_numpy(G[f'{pt}_w'])).sum().backward()
    assert torch.equal(x.grad, torch.from_numpy(G[f'{pt}_dx']))


def _channels_last(x, dev):
    return x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize('pt', TYPES)
def test_pooling_modes_match_reference(dev, pt):
    x = _channels_last(torch.from_numpy(G['x']), dev).requires_grad_(True)      # bf16-exact values
    m = T.POOLINGS.get('Pooling')(in_channels=16, pooling_type=pt).to(dev)
    assert m.out_channels == G[f'{pt}_y'].shape[1]
    y = m(x)
    want = torch.from_numpy(G[f'{pt}_y'])
    assert tuple(y.shape) == tuple(want.shape)
    assert float((y.float().cpu() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())      # one bf16 rounding
    # gradient weights on the bf16 grid so that the max route is exact: where the max went, the gradient is the weight
    w = torch.from_numpy(G[f'{pt}_w']).bfloat16().float()
    (y.float() * w.to(dev)).sum().backward()
    xr = torch.from_numpy(G['x']).clone().requires_grad_(True)
    (timm_min.SelectAdaptivePool2d(1, pt, flatten=True)(xr) * w).sum().backward()
    assert rel_err(x.grad.float().cpu(), xr.grad) < 4e-3
    if pt == 'max':
        # index-exact: the same pixels carry the gradient (ties -> the first maximal pixel), everything else is 0
        assert torch.equal(x.grad.float().cpu() != 0, xr.grad != 0)
        assert torch.equal(x.grad.float().cpu(), xr.grad.bfloat16().float())


@pytest.mark.parametrize('pt', TYPES)
def test_pooling_linear_matches_reference(dev, pt):
    m = T.POOLINGS.get('PoolingLinear')(in_channels=16, out_channels=24, pooling_type=pt)
    assert tuple(m.fc.weight.shape) == tuple(G[f'{pt}_lin_weight'].shape) and m.out_channels == 24
    with torch.no_grad():
        m.fc.weight.copy_(torch.from_numpy(G[f'{pt}_lin_weight']))
        m.fc.bias.copy_(torch.from_numpy(G[f'{pt}_lin_bias']))
    m.to(dev)
    x = _channels_last(torch.from_numpy(G['x']), dev).requires_grad_(True)
    z = m(x)
    wz = torch.from_numpy(G[f'{pt}_lin_wz'])
    (z.float() * wz.to(dev)).sum().backward()
    assert rel_err(z.float().cpu(), torch.from_numpy(G[f'{pt}_lin_z'])) < 1e-2
    assert rel_err(m.fc.weight.grad.cpu(), torch.from_numpy(G[f'{pt}_lin_dweight'])) < 1e-2
    assert rel_err(m.fc.bias.grad.cpu(), torch.from_numpy(G[f'{pt}_lin_dbias'])) < 1e-2
    assert rel_err(x.grad.float().cpu(), torch.from_numpy(G[f'{pt}_lin_dx'])) < 1.5e-2


def test_unknown_pool_type_raises():
    with pytest.raises(ValueError):
        T.POOLINGS.get('Pooling')(in_channels=16, pooling_type='median')
