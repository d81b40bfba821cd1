"""MetricsManager (reference metrics/metrics_manager.py:13-206) over the on-device Accuracy / F1Score statistics: values
against a plain numpy evaluation of the torchmetrics definitions (torchmetrics itself is third-party and absent), manager
semantics against the reference's documented behaviour.  Host stand-in and, marked gpu, libtok_gfx950.so."""
import numpy as np
import pytest
import torch

import torchok_amd as T
from helpers import cls_config
from torchok_amd.constructor.config import Phase
from torchok_amd.metrics import MetricsManager


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _reference(pred, tgt, c):
    tp = np.array([((pred == k) & (tgt == k)).sum() for k in range(c)], dtype=np.float64)
    pp = np.array([(pred == k).sum() for k in range(c)], dtype=np.float64)
    ap = np.array([(tgt == k).sum() for k in range(c)], dtype=np.float64)
    seen = (pp + ap) > 0
    f1 = 2 * tp / np.maximum(pp + ap, 1)
    acc = tp / np.maximum(ap, 1)
    return dict(acc_micro=tp.sum() / ap.sum(), acc_macro=acc[seen].mean(), f1_micro=2 * tp.sum() / (pp.sum() + ap.sum()),
                f1_macro=f1[seen].mean(), f1_weighted=(f1 * ap / ap.sum()).sum(), f1_none=f1)


def test_accuracy_and_f1_values(dev):
    g = torch.Generator().manual_seed(0)
    c = 7
    logits = torch.randn(500, c, generator=g)
    tgt = torch.randint(0, c - 1, (500,), generator=g)             # class 6 never occurs as a target
    logits[torch.arange(0, 500, 3), tgt[::3]] += 3.0                  # make a third of the rows right
    pred = logits.to(torch.bfloat16).float().argmax(1).numpy()
    want = _reference(pred, tgt.numpy(), c)
    mk = lambda name, **kw: T.METRICS.get(name)(task='multiclass', num_classes=c, **kw).to(dev)   # noqa: E731
    metrics = dict(acc_micro=mk('Accuracy'), acc_macro=mk('Accuracy', average='macro'), f1_micro=mk('F1Score'),
                   f1_macro=mk('F1Score', average='macro'), f1_weighted=mk('F1Score', average='weighted'),
                   f1_none=mk('F1Score', average='none'))
    for m in metrics.values():
        m.update(preds=logits[:300].to(dev), target=tgt[:300].to(dev))     # two updates accumulate
        m.update(preds=logits[300:].to(dev), target=tgt[300:].to(dev))
    for k, m in metrics.items():
        assert np.allclose(m.compute().cpu().numpy(), want[k], rtol=1e-6), k
    m = metrics['acc_micro']
    m.reset()
    m.update(preds=torch.from_numpy(pred).to(dev), target=tgt.to(dev))     # integer predictions
    assert np.allclose(float(m.compute()), want['acc_micro'], rtol=1e-6)
    ig = T.METRICS.get('Accuracy')(task='multiclass', num_classes=c, ignore_index=2).to(dev)
    ig.update(preds=logits.to(dev), target=tgt.to(dev))
    keep = tgt.numpy() != 2
    assert np.allclose(float(ig.compute()), (pred[keep] == tgt.numpy()[keep]).mean(), rtol=1e-6)
    with pytest.raises(NotImplementedError):
        T.METRICS.get('Accuracy')(task='binary')
    with pytest.raises(KeyError):
        T.METRICS.get('AUROC')


def test_precision_recall_fbeta_specificity(dev):
    """The StatScores family from the same on-device counts, against the torchmetrics multiclass definitions in numpy."""
    g = torch.Generator().manual_seed(4)
    c = 6
    logits = torch.randn(400, c, generator=g)
    tgt = torch.randint(0, c - 1, (400,), generator=g)             # class 5 never occurs as a target
    logits[:, 4] -= 50.0                                            # and class 4 is never predicted
    logits[torch.arange(0, 400, 2), tgt[::2]] += 3.0
    pred = logits.to(torch.bfloat16).float().argmax(1).numpy()
    t = tgt.numpy()
    tp = np.array([((pred == k) & (t == k)).sum() for k in range(c)], dtype=np.float64)
    pp = np.array([(pred == k).sum() for k in range(c)], dtype=np.float64)
    ap = np.array([(t == k).sum() for k in range(c)], dtype=np.float64)
    fp, fn = pp - tp, ap - tp
    tn = len(t) - pp - ap + tp
    seen = (pp + ap) > 0
    div = lambda a, b: np.where(b > 0, a / np.maximum(b, 1e-300), 0.0)      # noqa: E731
    per = dict(Precision=div(tp, tp + fp), Recall=div(tp, tp + fn), Specificity=div(tn, tn + fp),
               FBetaScore=div(1.25 * tp, 1.25 * tp + 0.25 * fn + fp))
    micro = dict(Precision=tp.sum() / (tp + fp).sum(), Recall=tp.sum() / (tp + fn).sum(), Specificity=tn.sum() / (tn + fp).sum(),
                 FBetaScore=1.25 * tp.sum() / (1.25 * tp.sum() + 0.25 * fn.sum() + fp.sum()))
    for name in per:
        kw = dict(beta=0.5) if name == 'FBetaScore' else {}
        for avg, want in (('micro', micro[name]), ('macro', per[name][seen].mean()), ('none', per[name]),
                          ('weighted', (per[name] * ap / ap.sum()).sum())):
            m = T.METRICS.get(name)(task='multiclass', num_classes=c, average=avg, **kw).to(dev)
            m.update(preds=logits[:150].to(dev), target=tgt[:150].to(dev))
            m.update(preds=logits[150:].to(dev), target=tgt[150:].to(dev))
            assert np.allclose(m.compute().cpu().numpy(), want, rtol=1e-6), (name, avg)
    assert abs(micro['Precision'] - (pred == t).mean()) < 1e-12        # single-label: micro precision = recall = accuracy


def test_confusion_matrix(dev):
    g = torch.Generator().manual_seed(6)
    c = 5
    logits = torch.randn(600, c, generator=g)
    tgt = torch.randint(0, c - 1, (600,), generator=g)             # class 4 never occurs as a target: an empty row
    pred = logits.to(torch.bfloat16).float().argmax(1)
    want = torch.bincount(tgt * c + pred, minlength=c * c).view(c, c)
    m = T.METRICS.get('ConfusionMatrix')(task='multiclass', num_classes=c).to(dev)
    m.update(preds=logits[:250].to(dev), target=tgt[:250].to(dev))
    m.update(preds=pred[250:].to(dev), target=tgt[250:].to(dev))    # label predictions accumulate into the same matrix
    assert torch.equal(m.compute().cpu(), want)                     # exact
    rows = T.METRICS.get('ConfusionMatrix')(task='multiclass', num_classes=c, normalize='true').to(dev)
    rows.update(preds=logits.to(dev), target=tgt.to(dev))
    got = rows.compute().cpu()
    assert torch.allclose(got[:4].sum(1), torch.ones(4)) and float(got[4].abs().sum()) == 0.0
    assert torch.allclose(got[:4], want[:4].float() / want[:4].sum(1, keepdim=True))
    ig = T.METRICS.get('ConfusionMatrix')(task='multiclass', num_classes=c, ignore_index=1).to(dev)
    ig.update(preds=logits.to(dev), target=tgt.to(dev))
    assert int(ig.compute()[1].sum()) == 0 and int(ig.compute().sum()) == int((tgt != 1).sum())
    m.reset()
    assert int(m.compute().sum()) == 0
    with pytest.raises(ValueError):
        T.METRICS.get('ConfusionMatrix')(task='multiclass', num_classes=c, normalize='rows')


def test_mean_absolute_and_squared_error(dev):
    g = torch.Generator().manual_seed(9)
    p = torch.randn(300, 4, generator=g).bfloat16()
    t = torch.randn(300, 4, generator=g)
    d = (p.float() - t).double()
    for name, kw, want in (('MeanAbsoluteError', {}, d.abs().mean()), ('MeanSquaredError', {}, d.pow(2).mean()),
                           ('MeanSquaredError', dict(squared=False), d.pow(2).mean().sqrt())):
        m = T.METRICS.get(name)(**kw).to(dev)
        m.update(preds=p[:100].to(dev), target=t[:100].to(dev))
        m.update(preds=p[100:].to(dev), target=t[100:].to(dev))
        assert abs(float(m.compute()) - float(want)) < 2e-6 * float(want), name
        m.reset()
        assert float(m.compute()) == 0.0
    with pytest.raises(RuntimeError):
        T.METRICS.get('MeanAbsoluteError')().to(dev).update(preds=p.to(dev), target=t[:, :2].to(dev))


def test_manager_semantics(dev):
    params = [dict(name='Accuracy', params=dict(task='multiclass', num_classes=4), mapping=dict(preds='prediction', target='target')),
              dict(name='F1Score', params=dict(task='multiclass', num_classes=4, average='macro'), tag='f1',
                   mapping=dict(preds='prediction', target='target'), phases=['VALID'], val_dataloader_idxs=[0, 1])]
    mm = MetricsManager(params).to(dev)
    assert len(mm.phase2metrics['TRAIN']) == 1 and len(mm.phase2metrics['VALID']) == 3
    pred = torch.eye(4)[[0, 1, 2, 3, 0, 1]].to(dev) * 5
    tgt = torch.tensor([0, 1, 2, 0, 0, 2]).to(dev)
    mm.update(Phase.TRAIN, prediction=pred, target=tgt, embeddings=None)
    mm.update(Phase.VALID, 1, prediction=pred, target=tgt)
    log = mm.on_epoch_end(Phase.TRAIN)
    assert set(log) == {'train/Accuracy'} and abs(float(log['train/Accuracy']) - 4 / 6) < 1e-6
    assert float(mm.on_epoch_end(Phase.TRAIN)['train/Accuracy']) == 0.0          # reset after the epoch
    vlog = mm.on_epoch_end(Phase.VALID)
    assert set(vlog) == {'valid/Accuracy', 'valid/f1_dataloader_0', 'valid/f1_dataloader_1'}
    assert float(vlog['valid/f1_dataloader_0']) == 0.0 and float(vlog['valid/f1_dataloader_1']) > 0.5
    with pytest.raises(ValueError, match='Cannot find'):
        mm.update(Phase.TRAIN, target=tgt)
    with pytest.raises(ValueError, match='identical names'):
        MetricsManager([params[0], params[0]])


def test_task_logs_configured_metrics(dev):
    cfg = cls_config('resnet18', 5)
    cfg['metrics'] = [dict(name='Accuracy', params=dict(task='multiclass', num_classes=5),
                           mapping=dict(preds='prediction', target='target'))]
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(dev).train()
    x, y = torch.randn(8, 3, 32, 32).to(dev), torch.randint(0, 5, (8,)).to(dev)
    task.training_step({'image': x, 'target': y}, 0)
    pred = task.forward_with_gt({'image': x, 'target': y})['prediction'].float().argmax(1)
    task.on_train_epoch_end()
    assert 'train/Accuracy' in task.logged
    task.eval()
    with torch.no_grad():
        task.validation_step({'image': x, 'target': y}, 0)
    task.on_validation_epoch_end()
    assert 0.0 <= float(task.logged['valid/Accuracy']) <= 1.0 and pred.shape == (8,)


def test_jaccard_index_and_legacy_forms(dev):
    g = torch.Generator().manual_seed(1)
    c = 4
    logits = torch.randn(2, c, 16, 16, generator=g)
    tgt = torch.randint(0, 3, (2, 16, 16), generator=g)                         # class 3 never a target
    pred = logits.to(torch.bfloat16).float().argmax(1).numpy().reshape(-1)
    t = tgt.numpy().reshape(-1)
    tp = np.array([((pred == k) & (t == k)).sum() for k in range(c)], dtype=np.float64)
    pp = np.array([(pred == k).sum() for k in range(c)], dtype=np.float64)
    ap = np.array([(t == k).sum() for k in range(c)], dtype=np.float64)
    iou = tp / (pp + ap - tp)
    # legacy form (segmentation_sweet_pepper.yaml:161-168): ignore_index drops class 0 from the mean, pixels all count
    m = T.METRICS.get('JaccardIndex')(num_classes=c, ignore_index=0).to(dev)
    m.update(preds=logits.to(dev), target=tgt.to(dev))
    assert np.allclose(float(m.compute()), iou[1:].mean(), rtol=1e-6)
    # task form: ignore_index drops the PIXELS labelled 0, macro over observed classes
    m2 = T.METRICS.get('JaccardIndex')(task='multiclass', num_classes=c, ignore_index=0).to(dev)
    m2.update(preds=logits.to(dev), target=tgt.to(dev))
    keep = t != 0
    tp2 = np.array([((pred[keep] == k) & (t[keep] == k)).sum() for k in range(c)], dtype=np.float64)
    un2 = np.array([(pred[keep] == k).sum() + (t[keep] == k).sum() for k in range(c)], dtype=np.float64) - tp2
    assert np.allclose(float(m2.compute()), (tp2[un2 > 0] / un2[un2 > 0]).mean(), rtol=1e-6)
    # legacy Accuracy(): the class count comes from the predictions
    a = T.METRICS.get('Accuracy')().to(dev)
    a.update(preds=logits.to(dev), target=tgt.to(dev))
    assert a.num_classes == c and np.allclose(float(a.compute()), (pred == t).mean(), rtol=1e-6)


def test_reference_metric_manager_known_answers():
    """tests/base_tests/metrics/metric_manager/test_metric_manager.py of the reference, case for case: user metrics
    registered in METRICS (plain modules with update / compute / reset instead of torchmetrics.Metric), 5 updates with
    `predict <- embedding`, `target <- target`; expected logs {'train/MockSumMetric': 5}, {'train/moc_sum': 5,
    'train/MockConstantMetric': 0}, {'train/MockDictMetric_target_shape': 10, 'train/MockDictMetric_embedding_size': 512},
    and an exception for a metric whose compute() cannot be called."""
    from torch import nn

    class _Mock(nn.Module):
        def __init__(self, **kw):
            super().__init__()
            self.reset()

        def update(self, predict, target):
            self.sum += 1

        def reset(self):
            self.sum = torch.tensor(0)

    class MockSumMetric(_Mock):
        def compute(self):
            return self.sum

    class MockConstantMetric(_Mock):
        def compute(self):
            return torch.tensor(0)

    class MockDictMetric(_Mock):
        def compute(self):
            return {'target_shape': torch.tensor(10), 'embedding_size': torch.tensor(512)}

    class MockRaiseMetric(_Mock):
        def compute(self, memory_block):
            return torch.tensor([1, 2])

    registered = []
    try:
        for cls in (MockSumMetric, MockConstantMetric, MockDictMetric, MockRaiseMetric):
            T.METRICS.register_class(cls)
            registered.append(cls.__name__)
        mapping = dict(predict='embedding', target='target')

        def run(names, tags):
            mm = MetricsManager([dict(name=n, mapping=mapping, tag=t, phases=[Phase.TRAIN]) for n, t in zip(names, tags)])
            for _ in range(5):
                mm.update(Phase.TRAIN, embedding=torch.rand(4, 512), target=torch.rand(4, 10))
            return {k: int(v) for k, v in mm.on_epoch_end(Phase.TRAIN).items()}
        assert run(['MockSumMetric'], [None]) == {'train/MockSumMetric': 5}
        assert run(['MockSumMetric', 'MockConstantMetric'], ['moc_sum', None]) == \
            {'train/moc_sum': 5, 'train/MockConstantMetric': 0}
        assert run(['MockDictMetric'], [None]) == {'train/MockDictMetric_target_shape': 10,
                                                   'train/MockDictMetric_embedding_size': 512}
        with pytest.raises(Exception):
            run(['MockRaiseMetric'], [None])
    finally:
        for n in registered:
            T.METRICS.entrypoints.pop(n, None)
