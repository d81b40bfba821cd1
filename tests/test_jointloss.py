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
