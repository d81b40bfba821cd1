"""Pin the oracle (CPU): oracle/torchok_ref.py must reproduce the golden vectors that
tests/golden/gen_golden.py produced by running the REFERENCE's own resnet.py / pooling.py /
classification_head.py / losses/base.py (on the restated timm subset) — inputs, logits, loss,
all parameter-gradient norms, small gradients in full, post-SGD-step parameters, BN running stats."""
import os

import numpy as np
import pytest
import torch

import oracle.torchok_ref as R
from helpers import deterministic_state

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('name', ['resnet18_cls_step', 'resnet50_cls_step'])
def test_oracle_reproduces_reference_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    torch.set_num_threads(4)
    backbone, classes, seed = str(g['backbone']), int(g['num_classes']), int(g['seed'])
    m = R.ClassificationModel(backbone, classes).train()
    m.load_state_dict(deterministic_state(m.state_dict(), seed))
    x, y = torch.from_numpy(g['x'].astype(np.float32)), torch.from_numpy(g['y'])
    feats = m.backbone.forward_features(x)
    assert [list(f.shape) for f in feats] == g['feat_shapes'].tolist()
    np.testing.assert_allclose([float(f.double().sum()) for f in feats], g['feat_sum'], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose([float((f.double() ** 2).sum()) for f in feats], g['feat_sumsq'], rtol=1e-5)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    out = m.forward_with_gt({'image': x, 'target': y})
    np.testing.assert_allclose(out['prediction'].detach().numpy(), g['prediction'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out['embeddings'].detach().numpy(), g['embeddings'], rtol=1e-4, atol=1e-5)
    loss = torch.nn.functional.cross_entropy(out['prediction'], y)
    assert abs(float(loss) - float(g['loss'])) < 1e-5
    loss.backward()
    names = [str(n) for n in g['param_names']]
    assert names == [n for n, _ in m.named_parameters()]
    gn = np.array([float(p.grad.double().norm()) for _, p in m.named_parameters()])
    np.testing.assert_allclose(gn, g['grad_norm'], rtol=2e-4)
    for n in g['small_names']:
        np.testing.assert_allclose(m.get_parameter(str(n)).grad.numpy(), g[f'grad__{n}'], rtol=1e-3, atol=1e-6)
    opt.step()
    pn = np.array([float(p.double().norm()) for _, p in m.named_parameters()])
    np.testing.assert_allclose(pn, g['post_step_norm'], rtol=1e-5)
    for n in g['small_names']:
        np.testing.assert_allclose(m.get_parameter(str(n)).detach().numpy(), g[f'post__{n}'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m.backbone.bn1.running_mean.numpy(), g['bn1_running_mean'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(m.backbone.bn1.running_var.numpy(), g['bn1_running_var'], rtol=1e-5)
    assert int(m.backbone.bn1.num_batches_tracked) == int(g['bn1_nbt']) == 2   # forward_features + forward_with_gt


def test_reference_shape_contracts():
    """tests/additional_tests/models/backbones/test_backbone.py:145-151 (resnet18 @ 2x3x64x64)."""
    m = R.resnet18()
    feats = m.forward_features(torch.rand(2, 3, 64, 64))
    assert [tuple(f.shape) for f in feats] == [(2, 3, 64, 64), (2, 64, 32, 32), (2, 64, 16, 16), (2, 128, 8, 8),
                                               (2, 256, 4, 4), (2, 512, 2, 2)]
    assert sum(p.numel() for p in R.ClassificationModel('resnet50', 1000).parameters()) == 25_557_032
    assert sum(p.numel() for p in R.ClassificationModel('resnet18', 1000).parameters()) == 11_689_512


def test_classification_head_golden():
    g = np.load(os.path.join(GOLD, 'classification_head.npz'))
    import torch.nn as nn
    fc = nn.Linear(32, 7)
    sd = deterministic_state({'fc.weight': fc.weight, 'fc.bias': fc.bias}, 3)
    y = torch.nn.functional.linear(torch.from_numpy(g['x'].astype(np.float32)), sd['fc.weight'], sd['fc.bias'])
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=1e-5, atol=1e-6)
    assert tuple(g['binary']) == (5,)      # num_classes == 1 squeezes the channel dim (classification_head.py:37-38)
