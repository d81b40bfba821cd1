"""Metrics: the MetricsManager of the reference (``torchok/metrics/metrics_manager.py:13-206``: per-phase metric lists,
mapping of task outputs to metric arguments, log names, epoch-end summary) over ON-DEVICE classification statistics.

The reference registers torchmetrics classes (third-party, absent here).  The two its classification recipes log every
training step (``examples/configs/classification_cifar10.yaml:134-150``) are provided with the torchmetrics constructor
surface they are configured through — ``Accuracy`` and ``F1Score`` for ``task='multiclass'`` (top-1; micro / macro /
weighted / none averages) — accumulating per-class {true positive, predicted, actual} counts with one kernel launch per
update and no host synchronisation until ``compute()``.  Other metric names resolve to a KeyError from the registry,
exactly like an unknown name in the reference."""
import numbers
from typing import Any, Dict, List, Optional

import torch
from torch import Tensor, nn

from . import _C
from .constructor import METRICS
from .constructor.config import Phase
from .engine.core import BF16, pad8, ptr, require_device, stream_ptr


def _sum_over_ranks(t: Tensor) -> Tensor:
    """torchmetrics `dist_reduce_fx='sum'`: the state every rank accumulated on its own shard is summed over the
    data-parallel group once, at compute() (one all-reduce per metric and epoch; identity outside a process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    t = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class _MulticlassStat(nn.Module):
    def __init__(self, task: str = None, num_classes: Optional[int] = None, average: Optional[str] = 'micro', top_k: int = 1,
                 ignore_index: Optional[int] = None, threshold: float = 0.5, multidim_average: str = 'global',
                 validate_args: bool = True, **kwargs):
        super().__init__()
        # task=None is the pre-0.11 torchmetrics call style the shipped recipes still use (torchmetrics 0.11.4 keeps it):
        # multiclass inferred from (N, C, ...) float predictions, num_classes taken from them when not given
        if task not in (None, 'multiclass') or top_k not in (1, None) or multidim_average != 'global':
            raise NotImplementedError(f"torchok_amd {type(self).__name__}: multiclass (task='multiclass' or the legacy "
                                      f"task-less form), top_k=1")
        if average not in ('micro', 'macro', 'weighted', 'none', None):
            raise ValueError(f'Expected argument `average` to be one of micro / macro / weighted / none, got {average}')
        self.legacy = task is None
        self.num_classes, self.average = (None if num_classes is None else int(num_classes)), average
        self.ignore_index = -100 if ignore_index is None else int(ignore_index)
        self.register_buffer('counts', torch.zeros(3, self.num_classes or 0, dtype=torch.int64), persistent=False)

    def update(self, preds: Tensor, target: Tensor) -> None:
        require_device(preds)
        if self.num_classes is None:
            if not preds.is_floating_point() or preds.dim() < 2:
                raise ValueError(f'{type(self).__name__}: give num_classes (it cannot be inferred from label predictions)')
            self.num_classes = int(preds.shape[1])
            self.counts = torch.zeros(3, self.num_classes, dtype=torch.int64, device=preds.device)
        c = self.num_classes
        tgt = target.reshape(-1).to(torch.int64).contiguous()
        if self.counts.device != preds.device:
            self.counts = self.counts.to(preds.device)
        lib, st = _C.lib(), stream_ptr()
        if preds.is_floating_point():
            if preds.dim() > 2:       # (N, C, ...) -> rows of C
                preds = preds.movedim(1, -1).reshape(-1, preds.shape[1])
            if preds.shape[-1] != c:
                raise ValueError(f'Expected {c} classes in `preds`, got shape {tuple(preds.shape)}')
            p = preds.detach()
            if p.dtype != BF16 or p.stride(-1) != 1:
                p = p.to(BF16).contiguous()
            _C.check(lib.tok_cls_stats_update(ptr(p), None, ptr(tgt), p.shape[0], c, p.stride(0), self._pixel_ignore(),
                                              ptr(self.counts), st), 'tok_cls_stats_update')
        else:
            lab = preds.detach().reshape(-1).to(torch.int64).contiguous()
            _C.check(lib.tok_cls_stats_update(None, ptr(lab), ptr(tgt), lab.shape[0], c, pad8(c), self._pixel_ignore(),
                                              ptr(self.counts), st), 'tok_cls_stats_update')

    def _pixel_ignore(self) -> int:
        return self.ignore_index

    def reset(self) -> None:
        self.counts.zero_()

    def _stats(self):
        counts = _sum_over_ranks(self.counts).to('cpu', torch.float64)     # the one host sync, at epoch end
        return counts[0], counts[1], counts[2]

    def _reduce(self, per_class: Tensor, support: Tensor, seen: Tensor) -> Tensor:
        if self.average in ('none', None):
            return per_class.float()
        if self.average == 'weighted':
            return (per_class * support / support.sum().clamp_min(1)).sum().float()
        return (per_class[seen].mean() if seen.any() else per_class.sum() * 0).float()   # macro over observed classes


@METRICS.register_class
class Accuracy(_MulticlassStat):
    def compute(self) -> Tensor:
        tp, pp, ap = self._stats()
        if self.average == 'micro':
            return (tp.sum() / ap.sum().clamp_min(1)).float()
        return self._reduce(tp / ap.clamp_min(1), ap, (ap + pp) > 0)


@METRICS.register_class
class F1Score(_MulticlassStat):
    def compute(self) -> Tensor:
        tp, pp, ap = self._stats()
        if self.average == 'micro':      # single-label multiclass: micro F1 == accuracy
            return (2 * tp.sum() / (pp.sum() + ap.sum()).clamp_min(1)).float()
        return self._reduce(2 * tp / (pp + ap).clamp_min(1), ap, (ap + pp) > 0)


class _FromCounts(_MulticlassStat):
    """Metrics of the torchmetrics StatScores family from the same {tp, predicted, actual} counts: fp = predicted - tp,
    fn = actual - tp, tn = rows - predicted - actual + tp; micro = the ratio of the summed terms, macro = the mean over
    the classes that occur in prediction or target, weighted = by support (torchmetrics 0.11 `_adjust_weights_safe_divide`)."""

    def _ratio(self, tp, fp, fn, tn):
        raise NotImplementedError

    def compute(self) -> Tensor:
        tp, pp, ap = self._stats()
        fp, fn = pp - tp, ap - tp
        tn = ap.sum() - pp - ap + tp
        if self.average == 'micro':
            num, den = self._ratio(tp.sum(), fp.sum(), fn.sum(), tn.sum())
            return (num / den.clamp_min(1)).float() if den > 0 else num.float() * 0
        num, den = self._ratio(tp, fp, fn, tn)
        return self._reduce(torch.where(den > 0, num / den.clamp_min(1e-300), torch.zeros_like(num)), ap, (ap + pp) > 0)


@METRICS.register_class
class Precision(_FromCounts):
    def _ratio(self, tp, fp, fn, tn):
        return tp, tp + fp


@METRICS.register_class
class Recall(_FromCounts):
    def _ratio(self, tp, fp, fn, tn):
        return tp, tp + fn


@METRICS.register_class
class Specificity(_FromCounts):
    def _ratio(self, tp, fp, fn, tn):
        return tn, tn + fp


@METRICS.register_class
class FBetaScore(_FromCounts):
    def __init__(self, task: str = None, beta: float = 1.0, **kwargs):
        super().__init__(task=task, **kwargs)
        if beta <= 0:
            raise ValueError(f'Expected argument `beta` to be a float larger than 0, but got {beta}.')
        self.beta = float(beta)

    def _ratio(self, tp, fp, fn, tn):
        b2 = self.beta ** 2
        return (1 + b2) * tp, (1 + b2) * tp + b2 * fn + fp


@METRICS.register_class
class JaccardIndex(_MulticlassStat):
    """Intersection over union per class from the same {tp, predicted, actual} counts.  Legacy (task-less) form, as in
    segmentation_sweet_pepper.yaml:161-168: `ignore_index` drops that CLASS from the average (every pixel still counts),
    classes absent from prediction and target score `absent_score`; task='multiclass': `ignore_index` drops the pixels."""

    def __init__(self, num_classes: int = None, average: Optional[str] = 'macro', ignore_index: Optional[int] = None,
                 absent_score: float = 0.0, threshold: float = 0.5, multilabel: bool = False,
                 reduction: str = 'elementwise_mean', task: str = None, **kwargs):
        if multilabel or reduction not in ('elementwise_mean', 'none', None):
            raise NotImplementedError('torchok_amd JaccardIndex: multiclass, mean / none reduction')
        super().__init__(task=task, num_classes=num_classes, average=average if task else 'macro',
                         ignore_index=ignore_index, **kwargs)
        self.absent_score, self.reduction, self.drop_class = absent_score, reduction, ignore_index

    def _pixel_ignore(self) -> int:
        return -100 if self.legacy else self.ignore_index

    def compute(self) -> Tensor:
        tp, pp, ap = self._stats()
        union = pp + ap - tp
        iou = torch.where(union > 0, tp / union.clamp_min(1), torch.full_like(tp, self.absent_score))
        if self.legacy:
            if self.drop_class is not None and 0 <= self.drop_class < iou.numel():
                iou = torch.cat([iou[:self.drop_class], iou[self.drop_class + 1:]])
            return iou.float() if self.reduction in ('none', None) else iou.mean().float()
        return self._reduce(iou, ap, union > 0)


@METRICS.register_class
class ConfusionMatrix(_MulticlassStat):
    """Multiclass confusion matrix [target][prediction] (metrics/__init__.py:53), accumulated on the device with exact int64
    atomics; `normalize` None / 'none' / 'true' (rows) / 'pred' (columns) / 'all' as in torchmetrics."""

    def __init__(self, task: str = None, num_classes: Optional[int] = None, normalize: Optional[str] = None,
                 ignore_index: Optional[int] = None, **kwargs):
        if normalize not in (None, 'none', 'true', 'pred', 'all'):
            raise ValueError(f'Argument `normalize` needs to be one of true / pred / all / none, got {normalize}')
        if num_classes is None:
            raise ValueError('ConfusionMatrix: `num_classes` is required')
        super().__init__(task=task, num_classes=num_classes, ignore_index=ignore_index, **kwargs)
        self.normalize = normalize
        self.register_buffer('confmat', torch.zeros(self.num_classes, self.num_classes, dtype=torch.int64), persistent=False)

    def update(self, preds: Tensor, target: Tensor) -> None:
        require_device(preds)
        c = self.num_classes
        tgt = target.reshape(-1).to(torch.int64).contiguous()
        if self.confmat.device != preds.device:
            self.confmat = self.confmat.to(preds.device)
        lib, st = _C.lib(), stream_ptr()
        if preds.is_floating_point():
            if preds.dim() > 2:
                preds = preds.movedim(1, -1).reshape(-1, preds.shape[1])
            if preds.shape[-1] != c:
                raise ValueError(f'Expected {c} classes in `preds`, got shape {tuple(preds.shape)}')
            p = preds.detach()
            if p.dtype != BF16 or p.stride(-1) != 1:
                p = p.to(BF16).contiguous()
            _C.check(lib.tok_confusion_update(ptr(p), None, ptr(tgt), p.shape[0], c, p.stride(0), self.ignore_index,
                                              ptr(self.confmat), st), 'tok_confusion_update')
        else:
            lab = preds.detach().reshape(-1).to(torch.int64).contiguous()
            _C.check(lib.tok_confusion_update(None, ptr(lab), ptr(tgt), lab.shape[0], c, pad8(c), self.ignore_index,
                                              ptr(self.confmat), st), 'tok_confusion_update')

    def reset(self) -> None:
        self.confmat.zero_()

    def compute(self) -> Tensor:
        cm = _sum_over_ranks(self.confmat)
        if self.normalize in (None, 'none'):
            return cm
        cm = cm.to(torch.float32)
        if self.normalize == 'true':
            cm = cm / cm.sum(1, keepdim=True)
        elif self.normalize == 'pred':
            cm = cm / cm.sum(0, keepdim=True)
        else:
            cm = cm / cm.sum()
        return torch.nan_to_num(cm, nan=0.0)      # torchmetrics: empty rows / columns read 0


class _ErrorSum(nn.Module):
    """MeanAbsoluteError / MeanSquaredError of the reference registry (metrics/__init__.py:76,78): running sum of the
    element errors and element count kept on the device (`tok_regression_loss_fwd`, 'sum'), no host sync until compute()."""
    kind = 0

    def __init__(self, **kwargs):
        super().__init__()
        self.register_buffer('total', torch.zeros(2, dtype=torch.float64), persistent=False)    # [sum, count]

    def update(self, preds: Tensor, target: Tensor) -> None:
        require_device(preds)
        if tuple(preds.shape) != tuple(target.shape):
            raise RuntimeError(f'Predictions and targets are expected to have the same shape, got {tuple(preds.shape)} and '
                               f'{tuple(target.shape)}')
        x = preds.detach().to(BF16).contiguous().view(-1)
        t = target.detach().to(torch.float32).contiguous().view(-1)
        buf = torch.empty(_C.TOK_CE_LOSS_FLOATS, dtype=torch.float32, device=x.device)
        _C.check(_C.lib().tok_regression_loss_fwd(ptr(x), ptr(t), x.numel(), self.kind, 1.0, 0, ptr(buf), stream_ptr()),
                 'tok_regression_loss_fwd')
        if self.total.device != x.device:
            self.total = self.total.to(x.device)
        self.total += buf[:2].double()

    def reset(self) -> None:
        self.total.zero_()

    def compute(self) -> Tensor:
        total = _sum_over_ranks(self.total)
        return (total[0] / total[1].clamp_min(1)).float()


@METRICS.register_class
class MeanAbsoluteError(_ErrorSum):
    kind = 0


@METRICS.register_class
class MeanSquaredError(_ErrorSum):
    kind = 1

    def __init__(self, squared: bool = True, **kwargs):
        super().__init__(**kwargs)
        self.squared = squared

    def compute(self) -> Tensor:
        mse = super().compute()
        return mse if self.squared else mse.sqrt()


class MetricWithUtils(nn.Module):
    """metrics_manager.py:13-75."""

    def __init__(self, metric: nn.Module, mapping: Dict[str, str], log_name: str, dataloader_idx: int):
        super().__init__()
        self.metric, self.mapping, self.log_name, self.dataloader_idx = metric, mapping, log_name, dataloader_idx

    def map_arguments(self, task_output: Dict[str, Any]) -> Dict[str, Any]:
        metric_input = {}
        for metric_target, metric_source in self.mapping.items():
            if metric_source not in task_output:
                raise ValueError(f'Cannot find {metric_source} for your mapping {metric_target} : {metric_source}. '
                                 f'You should either add {metric_source} output to your model or remove the mapping '
                                 f'from configuration')
            metric_input[metric_target] = task_output[metric_source]
        return metric_input

    def update(self, dataloader_idx: int = 0, **kwargs):
        if dataloader_idx == self.dataloader_idx:
            self.metric.update(**self.map_arguments(kwargs))

    def compute(self):
        return self.metric.compute()

    def reset(self):
        self.metric.reset()


def _phases(metric_params) -> List[Phase]:
    raw = metric_params.get('phases')
    if raw is None:
        return list(Phase)
    return [p if isinstance(p, Phase) else Phase[str(p).upper()] for p in raw]


class MetricsManager(nn.Module):
    """metrics_manager.py:78-206."""

    def __init__(self, params: List[dict]):
        super().__init__()
        self.phase2metrics = nn.ModuleDict()
        for phase in Phase:
            self.phase2metrics[phase.name] = self._get_phase_metrics(params or [], phase)

    @staticmethod
    def _get_phase_metrics(params: List[dict], phase: Phase) -> nn.ModuleList:
        added, metrics = [], []
        for mp in params:
            if phase not in _phases(mp):
                continue
            base = mp['name'] if mp.get('tag') is None else mp['tag']
            if phase == Phase.VALID:
                idxs = mp.get('val_dataloader_idxs') or [0]
            elif phase == Phase.TEST:
                idxs = mp.get('test_dataloader_idxs') or [0]
            else:
                idxs = [0]
            names = [f'{base}_dataloader_{i}' for i in idxs] if (phase in (Phase.VALID, Phase.TEST) and len(idxs) > 1) \
                else [base]
            for n in names:
                if n in added:
                    raise ValueError(f'Got two metrics with identical names: {n}. '
                                     f'Please, set different prefixes for identical metrics in the config file.')
                added.append(n)
            for i, n in zip(idxs, names):
                metric = METRICS.get(mp['name'])(**(mp.get('params') or {}))
                metrics.append(MetricWithUtils(metric=metric, mapping=dict(mp['mapping']), log_name=n, dataloader_idx=i))
        return nn.ModuleList(metrics)

    def update(self, phase: Phase, dataloader_idx: int = 0, **kwargs):
        for m in self.phase2metrics[Phase(phase).name if not isinstance(phase, Phase) else phase.name]:
            m.update(dataloader_idx, **kwargs)

    @staticmethod
    def is_number(num: Any) -> bool:
        if isinstance(num, Tensor):
            return num.dim() == 0
        return isinstance(num, numbers.Number)

    def on_epoch_end(self, phase: Phase) -> Dict[str, Tensor]:
        phase = phase if isinstance(phase, Phase) else Phase(phase)
        log = {}
        for m in self.phase2metrics[phase.name]:
            value = m.compute()
            if isinstance(value, dict):
                out = {f'{phase.value}/{m.log_name}_{k}': v for k, v in value.items() if self.is_number(v)}
                if not out:
                    raise ValueError(f'Metric manager on_epoch_end method. Metric {m.log_name}'
                                     f'return dict with has no numeric values.')
                log.update(out)
            elif self.is_number(value):
                log[f'{phase.value}/{m.log_name}'] = value
            else:
                raise ValueError(f'Metric manager on_epoch_end method. Metric {m.log_name} return no numeric value.')
            m.reset()
        return log
