"""NT_XentLoss / SimCLRTask and TripletMarginLoss / TripletLearnTask against
tests/golden/unsupervised_losses.npz (the reference's own unsupervised.py; torch's TripletMarginLoss, which is what the
reference registers), on the host stand-in and, marked gpu, through libtok_gfx950.so."""
import os

import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import rel_err
from torchok_amd.constructor.config import apply_schema

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'unsupervised_losses.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def test_nt_xent(dev):
    e1 = torch.from_numpy(GOLD['ntx_e1']).to(dev).to(torch.bfloat16).requires_grad_(True)
    e2 = torch.from_numpy(GOLD['ntx_e2']).to(dev).to(torch.bfloat16).requires_grad_(True)
    loss = T.LOSSES.get('NT_XentLoss')(temperature=0.2)(emb1=e1, emb2=e2)
    assert abs(float(loss) - float(GOLD['ntx_loss'])) < 1e-3 * float(GOLD['ntx_loss'])
    loss.backward()
    assert rel_err(e1.grad.float(), torch.from_numpy(GOLD['ntx_d1'])) < 1e-2
    assert rel_err(e2.grad.float(), torch.from_numpy(GOLD['ntx_d2'])) < 1e-2
    with pytest.raises(NotImplementedError):
        T.LOSSES.get('NT_XentLoss')()(e1, e2, emb_m=e1)
    # reduction='sum' (CrossEntropyLoss over the 2B rows of the similarity matrix, unsupervised.py:22,52) = mean * 2B
    rows = 2 * e1.shape[0]
    e1.grad = e2.grad = None
    ls = T.LOSSES.get('NT_XentLoss')(reduction='sum', temperature=0.2)(emb1=e1, emb2=e2)
    assert abs(float(ls.detach()) - rows * float(GOLD['ntx_loss'])) < 1e-3 * rows * float(GOLD['ntx_loss'])
    ls.backward()
    assert rel_err(e1.grad.float(), rows * torch.from_numpy(GOLD['ntx_d1'])) < 1e-2


@pytest.mark.parametrize('tag,kw', [('tri', dict(margin=1.0)), ('tri_swap', dict(margin=0.5, swap=True))])
def test_triplet_margin(dev, tag, kw):
    a, p, n = (torch.from_numpy(GOLD[k]).to(dev).to(torch.bfloat16).requires_grad_(True) for k in ('tri_a', 'tri_p', 'tri_n'))
    loss = T.LOSSES.get('TripletMarginLoss')(**kw)(anchor=a, positive=p, negative=n)
    assert abs(float(loss) - float(GOLD[tag + '_loss'])) < 1e-3 * float(GOLD[tag + '_loss'])
    loss.backward()
    for t, k in ((a, '_da'), (p, '_dp'), (n, '_dn')):
        assert rel_err(t.grad.float(), torch.from_numpy(GOLD[tag + k])) < 1e-2, k
    a.grad = None
    ls = T.LOSSES.get('TripletMarginLoss')(reduction='sum', **kw)(anchor=a, positive=p, negative=n)    # = mean * rows
    assert abs(float(ls.detach()) - a.shape[0] * float(GOLD[tag + '_loss'])) < 1e-3 * a.shape[0] * float(GOLD[tag + '_loss'])
    ls.backward()
    assert rel_err(a.grad.float(), a.shape[0] * torch.from_numpy(GOLD[tag + '_da'])) < 1e-2


def _cfg(task, loss, mapping, loss_params=None):
    return apply_schema({
        'task': {'name': task,
                 'params': {'backbone_name': 'resnet18',
                            # zero_init_last (the default) silences every residual branch at step 0
                            'backbone_params': {'pretrained': False, 'in_channels': 3, 'zero_init_last': False},
                            'pooling_name': 'Pooling', 'head_name': 'LinearHead',
                            'head_params': {'out_channels': 32, 'normalize': True},
                            'inputs': [{'shape': [3, 64, 64], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': loss, 'params': loss_params or {}, 'mapping': mapping}]},
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.05, 'momentum': 0.9}}}],
        'data': {}, 'trainer': {'precision': 'bf16'}})


def test_simclr_and_triplet_tasks(dev):
    """Two / three forwards of one model per step: every parameter receives the accumulated gradient and moves."""
    for task_name, loss, mapping, keys, lp in (
            ('SimCLRTask', 'NT_XentLoss', {'emb1': 'emb1', 'emb2': 'emb2'}, ('image_0', 'image_1'), {'temperature': 0.5}),
            ('TripletLearnTask', 'TripletMarginLoss', {'anchor': 'anchor', 'positive': 'positive', 'negative': 'negative'},
             ('anchor', 'positive', 'negative'), {})):
        cfg = _cfg(task_name, loss, mapping, lp)
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(dev).train()
        batch = {k: torch.randn(8, 3, 64, 64).to(dev) for k in keys}
        fw = task.forward_with_gt(batch)
        assert set(fw) == set(mapping.values()) and all(v.shape == (8, 32) for v in fw.values())
        out = task.training_step(batch, 0)
        opt = task.configure_optimizers()[0]['optimizer']
        opt.zero_grad()
        out['loss'].backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in task.parameters())
        assert int(task.backbone.bn1.num_batches_tracked) == 2 * len(keys)      # forward_with_gt + training_step
        before = {n: p.detach().clone() for n, p in task.named_parameters()}
        opt.step()
        assert all(not torch.equal(before[n], p.detach()) for n, p in task.named_parameters())
