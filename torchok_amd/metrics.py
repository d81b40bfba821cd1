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
        tp, pp, ap = (self.counts[i].to('cpu', torch.float64) for i in range(3))     # the one host sync, at epoch end
        return tp, pp, ap

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
