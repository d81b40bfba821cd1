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


@pytest.mark.parametrize('losses', ['ce+dice', 'dice', 'bce'])
def test_lazy_upsampled_logits_keep_every_loss_gradient(dev, losses, monkeypatch):
    """SegmentationHead returns a storage-less UpsampledLogits in training; `Function.apply` of Dice / BCE /
    regression saw a tensor without grad_fn and their gradient was silently dropped (CE + Dice trained on CE only).  The lazy
    path must give the gradients of the materialised path (TOK_FUSE_UPSAMPLE_CE=0) for the head AND the backbone."""
    from test_hrnet import seg_config
    from torchok_amd.losses import cross_entropy as CE
    classes = 3
    x = torch.randn(2, 3, 64, 64).to(dev)
    y = torch.randint(0, classes, (2, 64, 64)).to(dev)

    def run(lazy):
        monkeypatch.setattr(CE, 'FUSE_UPSAMPLE_CE', lazy)
        torch.manual_seed(3)
        cfg = seg_config('hrnet_w18_small', classes=classes, size=64)
        spec = type(cfg.joint_loss.losses[0])
        if losses == 'ce+dice':
            cfg.joint_loss.losses[0].params = {}
            cfg.joint_loss.losses.append(spec(name='DiceLoss', params={'mode': 'multiclass'},
                                              mapping={'input': 'prediction', 'target': 'target'}))
        elif losses == 'dice':
            cfg.joint_loss.losses[:] = [spec(name='DiceLoss', params={'mode': 'multiclass'},
                                             mapping={'input': 'prediction', 'target': 'target'})]
        else:
            cfg.joint_loss.losses[:] = [spec(name='BCEWithLogitsLoss', params={},
                                             mapping={'input': 'prediction', 'target': 'target'})]
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(dev).train()
        tgt = y if losses != 'bce' else torch.nn.functional.one_hot(y, classes).permute(0, 3, 1, 2).float()
        out = task.training_step({'image': x, 'target': tgt}, 0)
        if lazy:
            pred = task.forward_with_gt({'image': x, 'target': tgt})['prediction']
            assert isinstance(pred, CE.UpsampledLogits)
        assert out['loss'].grad_fn is not None
        out['loss'].backward()
        return float(out['loss'].detach()), {n: p.grad.detach().float().clone() for n, p in task.named_parameters()
                                            if p.grad is not None}
    l0, g0 = run(False)
    l1, g1 = run(True)
    assert abs(l0 - l1) < 2e-3 * max(1.0, abs(l0))
    assert g0.keys() == g1.keys() and len(g0) > 50
    for n in ('head.classifier.weight', 'head.classifier.bias'):
        assert float(g1[n].abs().max()) > 0
        assert rel_err(g1[n], g0[n]) < 2e-2, (n, rel_err(g1[n], g0[n]))
    stem = next(n for n in g0 if n.startswith('backbone.conv1'))
    assert rel_err(g1[stem], g0[stem]) < 6e-2, rel_err(g1[stem], g0[stem])


def test_upsampled_logits_expose_no_fake_memory(dev):
    """data_ptr / stride / is_contiguous / .data of the wrapper answer from the REAL tensor (the wrapper
    has no storage); shape / dtype / device queries do not materialise."""
    from torchok_amd.losses.cross_entropy import UpsampledLogits
    low = torch.randn(2, 8, 4, 4).to(dev).to(torch.bfloat16)
    u = UpsampledLogits(low, (16, 16))
    assert tuple(u.shape) == (2, 8, 16, 16) and u.dtype == low.dtype and u.dim() == 4 and u.numel() == 2 * 8 * 256
    assert u.device == low.device and u._full is None
    full = u.materialize()
    assert u.data_ptr() == full.data_ptr() != 0
    assert u.stride() == full.stride() and u.is_contiguous() == full.is_contiguous()
    assert u.data.data_ptr() == full.data_ptr()
