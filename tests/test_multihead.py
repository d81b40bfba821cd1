"""MultiHeadClassificationTask (reference tasks/multihead_classification.py:12-149): several named heads on one pooled
embedding, per-head targets and optional per-sample conditions.  The wiring is host code over units that have their own
parity tests, so the checks are structural and exact: a head of the multi-head task must reproduce the single-head
ClassificationTask bit for bit on the same weights, a condition must select exactly its rows, and the joint loss must
be the normalised sum of the per-head cross entropies with gradients reaching every head and the backbone."""
import pytest
import torch

import torchok_amd as T
from helpers import cls_config, copy_state, perturb_
from torchok_amd.constructor.config import apply_schema


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def multihead_config():
    heads = [{'type': 'ClassificationHead', 'name': 'colour', 'target': 'colour', 'params': {'num_classes': 10}},
             {'type': 'ClassificationHead', 'name': 'shape', 'target': 'kind', 'params': {'num_classes': 5}}]
    return apply_schema({
        'task': {'name': 'MultiHeadClassificationTask',
                 'params': {'backbone_name': 'resnet18', 'backbone_params': {'pretrained': False, 'in_channels': 3},
                            'pooling_name': 'Pooling', 'heads': heads,
                            'inputs': [{'shape': [3, 32, 32], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [
            {'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction_colour', 'target': 'target_colour'}},
            {'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction_shape', 'target': 'target_kind'}}]},
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.1, 'momentum': 0.9}}}],
        'data': {}, 'trainer': {'precision': 'bf16'}})


def test_heads_conditions_and_joint_loss(dev):
    torch.manual_seed(5)
    cfg = multihead_config()
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    perturb_(task, 3)
    single_cfg = cls_config('resnet18', 10)
    single = T.TASKS.get(single_cfg.task.name)(single_cfg, **single_cfg.task.params)
    copy_state(task.backbone, single.backbone)
    copy_state(task.heads['colour'], single.head)
    task, single = task.to(dev).train(), single.to(dev).train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 32, 32, generator=g).to(dev)
    batch = {'image': x, 'target_colour': torch.randint(0, 10, (16,), generator=g).to(dev),
             'target_kind': torch.randint(0, 5, (16,), generator=g).to(dev),
             'condition_kind': (torch.rand(16, generator=g) < 0.6).to(dev)}
    out = task.forward_with_gt(batch)
    ref = single.forward_with_gt({'image': x, 'target': batch['target_colour']})
    assert set(out) == {'embeddings', 'prediction_colour', 'target_colour', 'prediction_shape', 'target_kind'}
    assert torch.equal(out['embeddings'], ref['embeddings']) and torch.equal(out['prediction_colour'], ref['prediction'])
    n_sel = int(batch['condition_kind'].sum())
    assert 0 < n_sel < 16 and out['prediction_shape'].shape == (n_sel, 5) and out['target_kind'].shape == (n_sel,)
    assert torch.equal(out['target_kind'], batch['target_kind'][batch['condition_kind']])
    assert torch.equal(out['prediction_shape'], task.heads['shape'](out['embeddings'][batch['condition_kind']]))
    named = task(x)                                                       # forward(): a namedtuple over ALL samples
    assert named._fields == ('colour', 'shape') and named.shape.shape == (16, 5)
    step = task.training_step(batch, 0)
    ce = [torch.nn.functional.cross_entropy(out['prediction_colour'].float(), out['target_colour']),
          torch.nn.functional.cross_entropy(out['prediction_shape'].float(), out['target_kind'])]
    # JointLoss normalises the default weights (losses/base.py:43-54): 0.5 * CE_colour + 0.5 * CE_shape
    assert abs(float(step['loss'].detach()) - 0.5 * float((ce[0] + ce[1]).detach())) < 2e-3
    step['loss'].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in task.parameters())
    assert float(task.heads['shape'].fc.weight.grad.abs().sum()) > 0 and float(task.backbone.conv1.weight.grad.abs().sum()) > 0
    with pytest.raises(NotImplementedError):
        task.as_module()


def test_head_dropout(dev, monkeypatch):
    """LinearHead / ClassificationHead drop_rate > 0 (reference linear_head.py:27-28: F.dropout on the embedding before the
    Linear): with the Bernoulli draw pinned, output and gradients must be those of fc(x * scale); eval mode is the identity."""
    from helpers import rel_err
    torch.manual_seed(2)
    head = T.HEADS.get('ClassificationHead')(in_channels=64, num_classes=10, drop_rate=0.4).to(dev).train()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(32, 64, generator=g).to(dev).to(torch.bfloat16).requires_grad_(True)
    drawn = head.draw_dropout(4096, x.device)
    vals = drawn.unique().tolist()
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / 0.6) < 1e-6
    assert abs(float((drawn > 0).float().mean()) - 0.6) < 0.01
    scale = (torch.rand(32, 64, generator=g) < 0.6).float().div(0.6).to(dev)
    monkeypatch.setattr(type(head), 'draw_dropout', lambda self, rows, device: scale if self.training else None)
    y = head(x)
    w = torch.randn(32, 10, generator=g).to(dev)
    (y.float() * w).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    wr = head.fc.weight.detach().float().requires_grad_(True)
    yr = torch.nn.functional.linear((xr * scale).to(torch.bfloat16).float(), wr, head.fc.bias.detach().float())
    (yr * w).sum().backward()
    assert rel_err(y.float(), yr) < 1e-2 and rel_err(x.grad.float(), xr.grad) < 1e-2
    assert torch.all(x.grad[scale == 0] == 0)
    assert rel_err(head.fc.weight.grad.float(), wr.grad) < 1e-2
    plain = T.HEADS.get('ClassificationHead')(in_channels=64, num_classes=10).to(dev)
    plain.load_state_dict(head.state_dict())
    assert torch.equal(head.eval()(x.detach()), plain.eval()(x.detach()))
