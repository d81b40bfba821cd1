"""Metric-learning rows: ArcFaceHead, LinearHead(normalize), ContrastiveLoss,
PairwiseLearnTask against tests/golden/metric_heads.npz (outputs of the reference's own files,
tests/golden/gen_golden.py) and against oracle/metric_ref.py.

Every test runs twice: on the host stand-in (tests/fake_backend.py; checks the host logic on a CPU box)
and, marked gpu, through libtok_gfx950.so on the MI355X.  Tolerances are bf16 ones (activations and
gradients cross HBM as bf16; north_star: <= 1e-2 relative per tensor, looser where a cancellation
amplifies rounding — stated at the assert)."""
import os

import numpy as np
import pytest
import torch

import oracle.metric_ref as M
import torchok_amd as T
from helpers import deterministic_state, rel_err

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metric_heads.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def t(a, dev, dtype=None):
    x = torch.from_numpy(np.asarray(a))
    return (x if dtype is None else x.to(dtype)).to(dev)


def test_oracle_matches_reference_vectors():
    x, tg, w = (torch.from_numpy(GOLD[k]) for k in ('arc_x', 'arc_t', 'arc_w'))
    for tag, easy in (('arc', False), ('arc_easy', True)):
        wi, xi = w.clone().requires_grad_(True), x.clone().requires_grad_(True)
        y = M.arcface_forward(xi, wi, tg, float(GOLD[tag + '_margin']), float(GOLD[tag + '_scale']), easy)
        torch.nn.functional.cross_entropy(y, tg).backward()
        assert np.array_equal(y.detach().numpy(), GOLD[tag + '_out'])
        assert np.allclose(xi.grad.numpy(), GOLD[tag + '_dx'], rtol=1e-5, atol=1e-7)
        assert np.allclose(wi.grad.numpy(), GOLD[tag + '_dw'], rtol=1e-5, atol=1e-7)
        assert np.array_equal(M.arcface_forward(x, w, None, 0, 0, training=False).numpy(), GOLD[tag + '_eval'])
    assert M.arcface_defaults(64, 10) == (float(GOLD['arc_scale']), float(GOLD['arc_margin']))
    lab = torch.from_numpy(GOLD['con_lab'])
    assert np.array_equal(M.relevance_matrix(lab, 6).numpy(), GOLD['con_R'])
    e = torch.from_numpy(GOLD['con_e'])
    assert float(M.contrastive_loss(e, e, torch.from_numpy(GOLD['con_R']), 1.0)) == float(GOLD['con_loss'])


@pytest.mark.parametrize('tag,easy', [('arc', False), ('arc_easy', True)])
def test_arcface_head(dev, tag, easy):
    h = T.HEADS.get('ArcFaceHead')(64, 10, easy_margin=easy)
    assert h.scale == float(GOLD[tag + '_scale']) and h.margin == float(GOLD[tag + '_margin'])   # arcface_head.py:46-56
    with torch.no_grad():
        h.weight.copy_(torch.from_numpy(GOLD['arc_w']))
    h.to(dev).train()
    x = t(GOLD['arc_x'], dev).requires_grad_(True)
    tg = t(GOLD['arc_t'], dev)
    y = h(x, tg)
    assert y.shape == (16, 10)
    # logits are s * cos with s ~ 9.2: bf16 rounding of cos (2^-9 relative) -> 1e-2 of the logit scale
    assert rel_err(y.float(), torch.from_numpy(GOLD[tag + '_out'])) < 1e-2
    loss = torch.nn.functional.cross_entropy(y.float(), tg)
    assert abs(float(loss) - float(GOLD[tag + '_loss'])) < 2e-2 * float(GOLD[tag + '_loss'])
    loss.backward()
    assert rel_err(x.grad.float(), torch.from_numpy(GOLD[tag + '_dx'])) < 3e-2
    assert rel_err(h.weight.grad.float(), torch.from_numpy(GOLD[tag + '_dw'])) < 3e-2
    # eval: plain F.linear on the raw weight (arcface_head.py:120-121)
    ye = h.eval()(x.detach())
    assert rel_err(ye.float(), torch.from_numpy(GOLD[tag + '_eval'])) < 1e-2
    with pytest.raises(ValueError, match='Target is None'):
        h.train()(x.detach())


def test_arcface_reference_unit_tests(dev):
    """tests/additional_tests/models/heads/test_classification.py of the reference: shapes, (N, 1) targets."""
    h = T.HEADS.get('ArcFaceHead')(128, 10).to(dev)
    out = h(torch.rand(2, 128).to(dev), torch.tensor([[4], [8]]).to(dev))
    assert out.shape == (2, 10)
    assert h.weight.shape == (10, 128)
    with pytest.raises(ValueError):
        T.HEADS.get('ArcFaceHead')(128, 10, dynamic_margin=True)
    hd = T.HEADS.get('ArcFaceHead')(128, 10, dynamic_margin=True, num_warmup_steps=10, min_margin=0.1).to(dev)
    with pytest.raises(AttributeError):      # broken in the reference too
        hd(torch.rand(2, 128).to(dev), torch.tensor([4, 8]).to(dev))


def test_linear_head_normalize(dev):
    h = T.HEADS.get('LinearHead')(64, 24, normalize=True)
    h.load_state_dict(deterministic_state(h.state_dict(), 22))
    h.to(dev)
    x = t(GOLD['arc_x'], dev).requires_grad_(True)
    y = h(x)
    assert rel_err(y.float(), torch.from_numpy(GOLD['lin_out'])) < 1e-2
    assert torch.allclose(y.float().norm(dim=1).cpu(), torch.ones(16), atol=1e-2)
    (y.float() * torch.linspace(-1, 1, 24).to(dev)).sum().backward()
    assert rel_err(x.grad.float(), torch.from_numpy(GOLD['lin_dx'])) < 3e-2
    assert rel_err(h.fc.weight.grad, torch.from_numpy(GOLD['lin_dw'])) < 3e-2
    # d bias of a normalised output is a near-cancelling sum over the batch: judged against the gradient scale
    db = torch.from_numpy(GOLD['lin_db'])
    assert float((h.fc.bias.grad.cpu() - db).abs().max()) < 3e-2 * float(torch.from_numpy(GOLD['lin_dw']).abs().max()) * 8


def test_contrastive_loss_same_tensor(dev):
    """emb1 is emb2 — the way PairwiseLearnTask.forward_with_gt hands them to JointLoss (pairwise_task.py:79)."""
    e32 = t(GOLD['con_e'], dev)
    e = e32.to(torch.bfloat16).requires_grad_(True)
    R = t(GOLD['con_R'], dev)
    cl = T.LOSSES.get('ContrastiveLoss')(margin=1.0)
    loss = cl(emb1=e, emb2=e, R=R)
    # oracle evaluated at the bf16-rounded embeddings isolates kernel error from input rounding
    er = e.detach().float().cpu().requires_grad_(True)
    lo = M.contrastive_loss(er, er, R.cpu(), 1.0)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 1e-4 * float(lo)
    assert abs(float(loss) - float(GOLD['con_loss'])) < 2e-2 * float(GOLD['con_loss'])
    (loss * 1.0).backward()
    assert rel_err(e.grad.float(), er.grad) < 1e-2
    assert rel_err(e.grad.float(), torch.from_numpy(GOLD['con_de'])) < 3e-2


def test_contrastive_loss_memory_bank(dev):
    """emb2 != emb1 (cross-batch memory form, pairwise.py:88-99): separate gradients."""
    e1 = t(GOLD['con_e'], dev).to(torch.bfloat16).requires_grad_(True)
    e2 = t(GOLD['con_e2'], dev).to(torch.bfloat16).requires_grad_(True)
    cl = T.LOSSES.get('ContrastiveLoss')(margin=1.0)
    loss = cl(emb1=e1, emb2=e2, R=t(GOLD['con_R2'], dev))
    assert abs(float(loss) - float(GOLD['con_loss2'])) < 2e-2 * float(GOLD['con_loss2'])
    loss.backward()
    assert rel_err(e1.grad.float(), torch.from_numpy(GOLD['con_de1'])) < 3e-2
    assert rel_err(e2.grad.float(), torch.from_numpy(GOLD['con_de2'])) < 3e-2
    with pytest.raises(ValueError):
        T.LOSSES.get('ContrastiveLoss')(margin=1.0, reg='L3')
    with pytest.raises(ValueError):
        T.LOSSES.get('ContrastiveLoss')(margin=1.0, reduction='max')


@pytest.mark.parametrize('tag,kw', [('l1', dict(reg='L1')), ('l2', dict(reg='L2', eps=0.05)), ('sum', dict(reduction='sum')),
                                    ('l1sum', dict(reg='L1', reduction='sum', eps=0.01))])
def test_contrastive_regularisers_and_sum(dev, tag, kw):
    """BasePairwiseLoss.regularize / apply_reduction (pairwise.py:28-64) against the reference's own outputs + gradients."""
    e = t(GOLD['con_e'], dev).to(torch.bfloat16).requires_grad_(True)
    R = t(GOLD['con_R'], dev)
    loss = T.LOSSES.get('ContrastiveLoss')(margin=1.0, **kw)(emb1=e, emb2=e, R=R)
    er = e.detach().float().cpu().requires_grad_(True)          # oracle at the bf16-rounded embeddings: kernel error only
    lo = M.contrastive_loss(er, er, R.cpu(), 1.0, **kw)
    lo.backward()
    assert abs(float(loss.detach()) - float(lo)) < 1e-4 * float(lo)
    assert abs(float(loss.detach()) - float(GOLD[f'con_{tag}_loss'])) < 2e-2 * float(GOLD[f'con_{tag}_loss'])
    loss.backward()
    assert rel_err(e.grad.float(), er.grad) < 1e-2
    assert rel_err(e.grad.float(), torch.from_numpy(GOLD[f'con_{tag}_de'])) < 3e-2


def _pairwise_cfg():
    from torchok_amd.constructor.config import apply_schema
    return apply_schema({
        'task': {'name': 'PairwiseLearnTask',
                 'params': {'backbone_name': 'resnet18', 'backbone_params': {'pretrained': False, 'in_channels': 3},
                            'pooling_name': 'Pooling', 'head_name': 'LinearHead',
                            'head_params': {'out_channels': 32, 'normalize': True}, 'num_classes': 6,
                            'inputs': [{'shape': [3, 64, 64], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': 'ContrastiveLoss', 'params': {'margin': 0.5},
                                   'mapping': {'emb1': 'emb1', 'emb2': 'emb2', 'R': 'R'}}]},
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.05, 'momentum': 0.9}}}],
        'data': {}, 'trainer': {'precision': 'bf16'}})


def test_pairwise_task_step(dev):
    """Row a11 end to end: backbone -> pooling -> LinearHead(normalize) -> ContrastiveLoss over the task's R,
    vs the oracle wiring (same weights), then one optimizer step moves every parameter."""
    import oracle.torchok_ref as Rf
    cfg = _pairwise_cfg()
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 31)
    task.load_state_dict(sd, strict=False)
    task.to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 3, 64, 64, generator=g)
    lab = torch.randint(0, 6, (16,), generator=g)
    out = task.forward_with_gt({'image': x.to(dev), 'target': lab.to(dev)})
    assert set(out) == {'emb1', 'emb2', 'R', 'target'} and out['emb1'] is out['emb2']
    assert np.array_equal(out['R'].cpu().numpy(), M.relevance_matrix(lab, 6).numpy())      # exact
    assert out['R'].dtype == torch.float32
    # multi-label targets (pairwise_task.py:103-105): any shared class, exact; rows without a label match nothing
    ml = torch.from_numpy(GOLD['con_ml']).to(dev)
    assert np.array_equal(task.calc_relevance_matrix(ml).cpu().numpy(), GOLD['con_Rml'])
    assert np.array_equal(task.calc_relevance_matrix(ml.long()).cpu().numpy(), GOLD['con_Rml'])
    assert np.array_equal(M.relevance_matrix(torch.from_numpy(GOLD['con_ml']), 9).numpy(), GOLD['con_Rml'])

    ref = Rf.ClassificationModel('resnet18', 32).train()      # same backbone/pooling/fc wiring, 32-d output
    ref.load_state_dict(sd)
    emb = M.linear_head_forward(ref.pooling(ref.backbone(x)), ref.head.fc.weight, ref.head.fc.bias, True)
    lo = M.contrastive_loss(emb, emb, M.relevance_matrix(lab, 6), 0.5)
    lo.backward()
    assert rel_err(out['emb1'].float(), emb) < 3e-2

    step = task.training_step({'image': x.to(dev), 'target': lab.to(dev)}, 0)
    assert abs(float(step['loss']) - float(lo)) < 5e-2 * abs(float(lo)) + 1e-3
    step['loss'].backward()
    rp = dict(ref.named_parameters())
    errs = [rel_err(p.grad, rp[n].grad) for n, p in task.named_parameters()]
    # yardstick: torch's own bf16-autocast run of the oracle on the same weights (hinge terms switch on/off
    # under bf16 rounding of the embeddings, so the end-to-end gradient noise is far above 1e-2)
    import copy
    ref2 = copy.deepcopy(ref)
    ref2.zero_grad()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        f2 = ref2.pooling(ref2.backbone(x))
        emb2 = M.linear_head_forward(f2, ref2.head.fc.weight, ref2.head.fc.bias, True)
    M.contrastive_loss(emb2.float(), emb2.float(), M.relevance_matrix(lab, 6), 0.5).backward()
    yard = [rel_err(p.grad, rp[n].grad) for n, p in ref2.named_parameters()]
    assert np.median(errs) < 1.5 * np.median(yard) + 1e-2, (np.median(errs), np.median(yard))
    opt = task.configure_optimizers()[0]['optimizer']
    before = {n: p.detach().clone() for n, p in task.named_parameters()}
    opt.step()
    assert all(not torch.equal(before[n], p.detach()) for n, p in task.named_parameters())
