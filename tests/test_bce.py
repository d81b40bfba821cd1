"""BCEWithLogitsLoss with an ignore value against tests/golden/bce_loss.npz — losses and input gradients of the reference's
own losses/classification/binary_cross_entropy.py — for the oracle restatement (CPU), the host stand-in and, marked gpu,
libtok_gfx950.so."""
import os

import numpy as np
import pytest
import torch

import oracle.bce_ref as O
import torchok_amd as T
from helpers import rel_err

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bce_loss.npz'))
CASES = [('a', 'x', 't'), ('b', 'x4', 't4')]


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def test_oracle_matches_reference_outputs():
    for tag, xk, tk in CASES:
        for red in ('mean', 'sum'):
            loss, grad = O.bce_with_logits_ignore(GOLD[xk], GOLD[tk], -1, red)
            assert abs(loss - float(GOLD[f'{tag}_{red}_loss'])) < 2e-6 * abs(loss)          # the reference sums in fp32
            assert np.abs(grad - GOLD[f'{tag}_{red}_dx']).max() < 1e-6 * max(1.0, np.abs(grad).max())
    loss, grad = O.bce_with_logits_ignore(GOLD['x'], np.maximum(GOLD['t'], 0), 0, 'mean')
    assert abs(loss - float(GOLD['ign0_loss'])) < 2e-6 * abs(loss) and np.abs(grad - GOLD['ign0_dx']).max() < 1e-7
    assert O.bce_with_logits_ignore(GOLD['x'], np.full_like(GOLD['t'], -1))[0] == float(GOLD['empty_loss']) == 0.0


@pytest.mark.parametrize('tag,xk,tk', CASES)
@pytest.mark.parametrize('red', ['mean', 'sum'])
def test_loss_and_gradient(dev, tag, xk, tk, red):
    x = torch.from_numpy(GOLD[xk]).to(dev).to(torch.bfloat16).requires_grad_(True)     # the golden logits are bf16-exact
    t = torch.from_numpy(GOLD[tk]).to(dev)
    loss = T.LOSSES.get('BCEWithLogitsLoss')(reduction=red)(x, t)
    ref = float(GOLD[f'{tag}_{red}_loss'])
    assert abs(float(loss.detach()) - ref) < 1e-5 * abs(ref)                     # fp32 elements, fp64 fold
    loss.backward()
    assert x.grad.shape == x.shape
    assert rel_err(x.grad.float(), torch.from_numpy(GOLD[f'{tag}_{red}_dx'])) < 4e-3   # bf16 gradient storage
    assert torch.all(x.grad[t == -1] == 0)


def test_ignore_value_empty_selection_and_errors(dev):
    x = torch.from_numpy(GOLD['x']).to(dev).to(torch.bfloat16).requires_grad_(True)
    t = torch.from_numpy(GOLD['t']).to(dev)
    loss = T.LOSSES.get('BCEWithLogitsLoss')(ignore_index=0)(x, t.clamp_min(0))
    assert abs(float(loss.detach()) - float(GOLD["ign0_loss"])) < 1e-5 * float(GOLD['ign0_loss'])
    loss.backward()
    assert rel_err(x.grad.float(), torch.from_numpy(GOLD['ign0_dx'])) < 4e-3
    x.grad = None
    empty = T.LOSSES.get('BCEWithLogitsLoss')()(x, torch.full_like(t, -1))
    assert float(empty.detach()) == 0.0
    empty.backward()
    assert torch.count_nonzero(x.grad) == 0
    # integer multi-hot targets and fp32 logits go through the same path (target.float(), :51)
    l32 = T.LOSSES.get('BCEWithLogitsLoss')()(torch.from_numpy(GOLD['x']).to(dev), t.long())
    assert abs(float(l32) - float(GOLD['a_mean_loss'])) < 1e-5 * float(GOLD['a_mean_loss'])
    with pytest.raises(NotImplementedError):
        T.LOSSES.get('BCEWithLogitsLoss')(pos_weight=[1.0, 2.0])
    with pytest.raises(NotImplementedError):
        T.LOSSES.get('BCEWithLogitsLoss')(reduction='none')
    with pytest.raises(ValueError):
        T.LOSSES.get('BCEWithLogitsLoss')()(x, t[:, :5])


@pytest.mark.parametrize('name,kw', [('L1Loss', {}), ('MSELoss', dict(reduction='sum')), ('SmoothL1Loss', dict(beta=0.7)),
                                     ('SmoothL1Loss', dict(beta=0.0)), ('HuberLoss', dict(delta=1.5)),
                                     ('HuberLoss', dict(delta=0.4, reduction='sum'))])
def test_regression_losses_vs_torch(dev, name, kw):
    """L1 / MSE / SmoothL1 / Huber are torch's own classes in the reference registry (losses/__init__.py:13-25): the oracle
    is torch.nn itself, evaluated on the bf16-exact prediction."""
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(6, 5, 9, generator=g) * 2).bfloat16()
    t = torch.randn(6, 5, 9, generator=g)
    xr = x.float().requires_grad_(True)
    ref = getattr(torch.nn, name)(**kw)(xr, t)
    ref.backward()
    xd = x.to(dev).requires_grad_(True)
    loss = T.LOSSES.get(name)(**kw)(xd, t.to(dev))
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6 * abs(float(ref.detach()))
    loss.backward()
    assert xd.grad.shape == x.shape and rel_err(xd.grad.float(), xr.grad) < 4e-3        # bf16 gradient storage
    with pytest.raises(ValueError):
        T.LOSSES.get(name)(**kw)(xd, t[:, :2].to(dev))
