"""JointLoss known answers — the one numeric test the reference holds on the hot path
(tests/base_tests/losses/test_base_losses.py:19-77: 8.0 / 10.0 / 5.0 / 15.0 / ValueError),
run against BOTH the build's JointLoss and the oracle's restatement."""
import pytest
import torch
from torch.nn import Module

import oracle.torchok_ref as R
from torchok_amd.losses import JointLoss


class Loss1(Module):
    def forward(self, input, target):
        return torch.abs(input * 10. - target)


class Loss2(Module):
    def forward(self, input, target):
        return torch.abs(input * 20. - target)


MAPS = [{'input': 'x', 'target': 'y'}] * 2
X, Y = torch.ones(1), torch.full((1,), 5.)


def test_weighted():
    total, tagged = JointLoss([Loss1(), Loss2()], MAPS, ['loss1', 'loss2'], [0.7, 0.3])(x=X, y=Y)
    torch.testing.assert_close(total, torch.tensor([8.]))
    torch.testing.assert_close(tagged['loss1'], torch.tensor([5.]))
    torch.testing.assert_close(tagged['loss2'], torch.tensor([15.]))
    ototal, otag = R.joint_loss([Loss1(), Loss2()], MAPS, ['loss1', 'loss2'], [0.7, 0.3], True, x=X, y=Y)
    torch.testing.assert_close(ototal, torch.tensor([8.]))
    torch.testing.assert_close(otag['loss2'], torch.tensor([15.]))


def test_unweighted_is_normalised():
    total, _ = JointLoss([Loss1(), Loss2()], MAPS, ['loss1', 'loss2'], [None, None])(x=X, y=Y)
    torch.testing.assert_close(total, torch.tensor([10.]))
    ototal, _ = R.joint_loss([Loss1(), Loss2()], MAPS, ['a', 'b'], [None, None], True, x=X, y=Y)
    torch.testing.assert_close(ototal, torch.tensor([10.]))


def test_partial_weights_raise():
    with pytest.raises(ValueError):
        JointLoss([Loss1(), Loss2()], MAPS, ['loss1', 'loss2'], [0.7, None])
    with pytest.raises(ValueError):
        R.joint_loss([Loss1(), Loss2()], MAPS, ['a', 'b'], [0.7, None], True, x=X, y=Y)


def test_tag_access_and_missing_mapping():
    jl = JointLoss([Loss1(), Loss2()], MAPS, ['loss1', None], [0.7, 0.3])
    assert isinstance(jl['loss1'], Loss1)
    with pytest.raises(KeyError):
        jl['loss2']
    with pytest.raises(ValueError):
        jl(x=X)            # 'y' missing from the model outputs (losses/base.py:110-112)


@pytest.mark.parametrize('shape', [(33, 10), (3, 5, 6, 7)])
@pytest.mark.parametrize('kw', [dict(label_smoothing=0.1), dict(reduction='sum'), dict(label_smoothing=0.2, reduction='sum',
                                                                                      ignore_index=2)])
def test_cross_entropy_label_smoothing_and_sum(fake_backend, shape, kw):
    """CrossEntropyLoss(label_smoothing, reduction='sum') against torch.nn.CrossEntropyLoss — the class the reference
    registers under this name (losses/__init__.py:26) — on bf16-exact logits: loss <= 1e-5, gradient = bf16 storage."""
    _ce_options_case('cpu', shape, kw)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(33, 10), (3, 5, 6, 7), (4096, 1000)])
@pytest.mark.parametrize('kw', [dict(label_smoothing=0.1), dict(reduction='sum'), dict(label_smoothing=0.2, reduction='sum',
                                                                                      ignore_index=2)])
def test_cross_entropy_label_smoothing_and_sum_gpu(shape, kw):
    _ce_options_case('cuda', shape, kw)


def _ce_options_case(dev, shape, kw):
    import torchok_amd as T
    g = torch.Generator().manual_seed(len(shape))
    x = (torch.randn(*shape, generator=g) * 2).bfloat16()
    classes = shape[1]
    t = torch.randint(0, classes, (shape[0],) + tuple(shape[2:]), generator=g)
    xr = x.float().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(**kw)(xr, t)
    ref.backward()
    xd = x.to(dev).requires_grad_(True)
    loss = T.LOSSES.get('CrossEntropyLoss')(**kw)(xd, t.to(dev))
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * abs(float(ref.detach()))
    loss.backward()
    err = (xd.grad.float().cpu() - xr.grad).abs().max() / xr.grad.abs().max()
    assert float(err) < 5e-3
