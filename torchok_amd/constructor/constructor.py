"""Factories for the hot-path components (reference ``torchok/constructor/constructor.py``):
optimizers incl. the mmcv-style ``paramwise_cfg`` (:86-251), schedulers (:254-262), JointLoss
(:367-382).  Data loaders / transforms / full MetricsManager are outside the hot-path scope."""
from typing import Any, Dict, List, Optional, Union

import torch
from torch import nn
from torch.nn import GroupNorm, LayerNorm, Module, ModuleList
from torch.nn.modules.batchnorm import _BatchNorm
from torch.nn.modules.instancenorm import _InstanceNorm
from torch.optim import Optimizer

from . import LOSSES, OPTIMIZERS, SCHEDULERS
from ..losses.base import JointLoss
from ..metrics import MetricsManager


class _ParamwiseRules:
    """Per-parameter learning-rate / weight-decay overrides.

    Precedence, as in the reference: the first `custom_keys` entry (longest key first, ties alphabetical) that is a
    substring of '<module path>.<parameter name>' decides alone; otherwise the bias learning-rate multiplier applies to
    non-norm biases and exactly one weight-decay multiplier applies: norm layer > depth-wise conv > bias."""
    NORMS = (_BatchNorm, _InstanceNorm, GroupNorm, LayerNorm)

    def __init__(self, optimizer_cfg: Dict, cfg: Dict):
        self.lr, self.wd = optimizer_cfg.get('lr', None), optimizer_cfg.get('weight_decay', None)
        custom = cfg.get('custom_keys', {})
        self.custom = [(k, custom[k]) for k in sorted(sorted(custom), key=len, reverse=True)]
        self.bias_lr = cfg.get('bias_lr_mult', 1.)
        self.decay = {'norm': cfg.get('norm_decay_mult', 1.), 'dwconv': cfg.get('dwconv_decay_mult', 1.),
                      'bias': cfg.get('bias_decay_mult', 1.)}

    def module_kind(self, module: nn.Module) -> str:
        if isinstance(module, self.NORMS):
            return 'norm'
        if isinstance(module, torch.nn.Conv2d) and module.in_channels == module.groups:
            return 'dwconv'
        return 'plain'

    def options(self, kind: str, path: str, name: str) -> Dict[str, float]:
        full = f'{path}.{name}'
        for key, mult in self.custom:
            if key in full:
                opts = {'lr': self.lr * mult.get('lr_mult', 1.)}
                if self.wd is not None:
                    opts['weight_decay'] = self.wd * mult.get('decay_mult', 1.)
                return opts
        opts = {}
        is_bias = name == 'bias'
        if is_bias and kind != 'norm':
            opts['lr'] = self.lr * self.bias_lr
        if self.wd is not None:
            which = kind if kind in ('norm', 'dwconv') else ('bias' if is_bias else None)
            if which is not None:
                opts['weight_decay'] = self.wd * self.decay[which]
        return opts


class Constructor:
    def __init__(self, hparams):
        self._hparams = hparams

    def configure_optimizers(self, modules: Union[Module, List[Module]], optim_idx: int = -1):
        optims_params = self._hparams.optimization
        if 0 <= optim_idx < len(optims_params):
            optims_params = [optims_params[optim_idx]]
        elif optim_idx >= len(optims_params):
            raise ValueError(f'You requested optimization with index {optim_idx} while '
                             f'there\'re only {len(optims_params)} optimization parameters are specified')
        opt_sched_list = []
        for optim_params in optims_params:
            optimizer = self.create_optimizer(modules, optim_params.optimizer)
            opt_sched = {'optimizer': optimizer}
            if optim_params.scheduler is not None:
                opt_sched['lr_scheduler'] = self._create_scheduler(optimizer, optim_params.scheduler)
            opt_sched_list.append(opt_sched)
        return opt_sched_list

    @staticmethod
    def create_optimizer(modules: Union[Module, List[Module]], optimizer_params) -> Optimizer:
        optimizer_class = OPTIMIZERS.get(optimizer_params.name)
        paramwise_cfg = optimizer_params.get('paramwise_cfg')
        optimizer_cfg = dict(optimizer_params.get('params') or {})
        if isinstance(modules, (tuple, list)):
            modules = ModuleList(modules)
        if not paramwise_cfg:
            # ONE group, registration order (reference :151-152; `no_weight_decay()` is never consulted)
            parameters = list(modules.parameters())
        else:
            parameters = []
            Constructor.add_params(parameters, modules, optimizer_cfg, paramwise_cfg)
        return optimizer_class(parameters, **optimizer_cfg)

    @staticmethod
    def add_params(parameters: List[Dict], module: nn.Module, optimizer_cfg: Dict,
                   paramwise_cfg: Optional[Dict] = None, prefix: str = '', is_dcn_module=None) -> None:
        """One param group per parameter, options from the mmcv-style `paramwise_cfg` (behaviour of reference :163-251).

        Walk order = the reference's recursion: a module's own parameters, then its children in registration order.
        `is_dcn_module` is accepted for signature compatibility; the reference only ever passes a falsy value down, so
        its `dcn_offset_lr_mult` rule can never fire and is not evaluated here."""
        rules = _ParamwiseRules(optimizer_cfg, paramwise_cfg or {})
        stack = [(prefix, module)]
        while stack:
            path, mod = stack.pop()
            kind = rules.module_kind(mod)
            for name, param in mod.named_parameters(recurse=False):
                group = {'params': [param]}
                if param.requires_grad:
                    group.update(rules.options(kind, path, name))
                parameters.append(group)
            children = [(f'{path}.{cn}' if path else cn, cm) for cn, cm in mod.named_children()]
            stack.extend(reversed(children))

    @staticmethod
    def _create_scheduler(optimizer: Optimizer, scheduler_params) -> Dict[str, Any]:
        scheduler_class = SCHEDULERS.get(scheduler_params.name)
        scheduler = scheduler_class(optimizer, **(scheduler_params.get('params') or {}))
        return {'scheduler': scheduler, **(scheduler_params.get('pl_params') or {})}

    def configure_metrics_manager(self):
        return MetricsManager(self._hparams.get('metrics') or [])

    def configure_losses(self) -> JointLoss:
        loss_modules, mappings, tags, weights = [], [], [], []
        for loss_config in self._hparams.joint_loss.losses:
            loss_modules.append(LOSSES.get(loss_config.name)(**(loss_config.get('params') or {})))
            mappings.append(loss_config.mapping)
            tags.append(loss_config.get('tag'))
            weights.append(loss_config.get('weight'))
        return JointLoss(loss_modules, mappings, tags, weights, self._hparams.joint_loss.normalize_weights)

    @property
    def hparams(self):
        return self._hparams
