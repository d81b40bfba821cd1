"""DiceLoss against tests/golden/dice_loss.npz — losses and input gradients of the reference's own
losses/segmentation/dice.py — on the host stand-in and, marked gpu, through libtok_gfx950.so; plus the shipped HRNet
recipe's loss pair (CrossEntropyLoss + DiceLoss on the same prediction, segmentation_sweet_pepper.yaml:16-27)."""
import os

import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import rel_err

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'dice_loss.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


@pytest.mark.parametrize('tag,kw', [('mc', {}), ('mc_log', dict(log_loss=True, smooth=1.0)), ('mc_sel', dict(classes=[0, 2, 4]))])
def test_multiclass(dev, tag, kw):
    z = torch.from_numpy(GOLD['z']).to(dev).to(torch.bfloat16).requires_grad_(True)
    t = torch.from_numpy(GOLD['t']).to(dev)
    loss = T.LOSSES.get('DiceLoss')('multiclass', **kw)(z, t)
    assert abs(float(loss) - float(GOLD[tag + '_loss'])) < 2e-3 * abs(float(GOLD[tag + '_loss'])) + 1e-5
    loss.backward()
    assert rel_err(z.grad.float(), torch.from_numpy(GOLD[tag + '_dz'])) < 1e-2     # bf16 gradient storage


@pytest.mark.parametrize('tag,kw', [('ml', dict(smooth=0.5)), ('ml_log_sel', dict(log_loss=True, smooth=1.0, classes=[1, 3, 4]))])
def test_multilabel(dev, tag, kw):
    """'multilabel' (dice.py:143,166-168): a sigmoid per class against dense (N, C, H, W) targets; the empty class is masked."""
    z = torch.from_numpy(GOLD['z']).to(dev).to(torch.bfloat16).requires_grad_(True)
    t = torch.from_numpy(GOLD['tm']).to(dev)
    loss = T.LOSSES.get('DiceLoss')('multilabel', **kw)(z, t)
    assert abs(float(loss.detach()) - float(GOLD[tag + '_loss'])) < 2e-3 * abs(float(GOLD[tag + '_loss'])) + 1e-5
    loss.backward()
    assert rel_err(z.grad.float(), torch.from_numpy(GOLD[tag + '_dz'])) < 1e-2     # bf16 gradient storage
    assert float(z.grad[:, 3].abs().max()) == 0.0                                  # no true pixel: no gradient
    with pytest.raises(ValueError):
        T.LOSSES.get('DiceLoss')('multilabel')(z.detach(), t[:, :4])


def test_binary_and_errors(dev):
    z = torch.from_numpy(GOLD['zb']).to(dev).to(torch.bfloat16).requires_grad_(True)
    t = torch.from_numpy(GOLD['tb']).to(dev)
    loss = T.LOSSES.get('DiceLoss')('binary', smooth=0.5)(z, t)
    assert abs(float(loss) - float(GOLD['bin_loss'])) < 2e-3 * abs(float(GOLD['bin_loss']))
    loss.backward()
    assert rel_err(z.grad.float(), torch.from_numpy(GOLD['bin_dz'])) < 1e-2
    with pytest.raises(ValueError):
        T.LOSSES.get('DiceLoss')('binary', classes=[0])
    with pytest.raises(ValueError):
        T.LOSSES.get('DiceLoss')('nope')
    with pytest.raises(ValueError):
        T.LOSSES.get('DiceLoss')('multiclass')(z.detach()[:, None].expand(-1, 3, -1, -1), t[:, :4])


def test_shipped_recipe_loss_pair(dev):
    """JointLoss(CrossEntropyLoss, DiceLoss(multiclass)) on one prediction: both gradients reach the segmentation head."""
    from test_hrnet import seg_config
    cfg = seg_config('hrnet_w18_small', classes=3, size=64)
    cfg.joint_loss.losses.append(type(cfg.joint_loss.losses[0])(name='DiceLoss', params={'mode': 'multiclass'},
                                                                 mapping={'input': 'prediction', 'target': 'target'}))
    cfg.joint_loss.losses[0].params = {}
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(dev).train()
    x = torch.randn(2, 3, 64, 64).to(dev)
    y = torch.randint(0, 3, (2, 64, 64)).to(dev)
    out = task.training_step({'image': x, 'target': y}, 0)
    fw = task.forward_with_gt({'image': x, 'target': y})
    ce = torch.nn.functional.cross_entropy(fw['prediction'].float(), y)
    dice = T.LOSSES.get('DiceLoss')('multiclass')(fw['prediction'], y)
    # JointLoss normalises the default weights (losses/base.py:43-54): 0.5 * CE + 0.5 * Dice
    assert abs(float(out['loss'].detach()) - 0.5 * (float(ce.detach()) + float(dice.detach()))) < 2e-2
    out['loss'].backward()
    assert all(p.grad is not None for p in task.head.parameters())
