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
        """mmcv-style per-parameter groups (reference :163-251)."""
        paramwise_cfg = paramwise_cfg or {}
        base_lr = optimizer_cfg.get('lr', None)
        base_wd = optimizer_cfg.get('weight_decay', None)
        custom_keys = paramwise_cfg.get('custom_keys', {})
        sorted_keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
        bias_lr_mult = paramwise_cfg.get('bias_lr_mult', 1.)
        bias_decay_mult = paramwise_cfg.get('bias_decay_mult', 1.)
        norm_decay_mult = paramwise_cfg.get('norm_decay_mult', 1.)
        dwconv_decay_mult = paramwise_cfg.get('dwconv_decay_mult', 1.)
        dcn_offset_lr_mult = paramwise_cfg.get('dcn_offset_lr_mult', 1.)
        is_norm = isinstance(module, (_BatchNorm, _InstanceNorm, GroupNorm, LayerNorm))
        is_dwconv = isinstance(module, torch.nn.Conv2d) and module.in_channels == module.groups
        for name, param in module.named_parameters(recurse=False):
            param_group = {'params': [param]}
            if not param.requires_grad:
                parameters.append(param_group)
                continue
            is_custom = False
            for key in sorted_keys:
                if key in f'{prefix}.{name}':
                    is_custom = True
                    param_group['lr'] = base_lr * custom_keys[key].get('lr_mult', 1.)
                    if base_wd is not None:
                        param_group['weight_decay'] = base_wd * custom_keys[key].get('decay_mult', 1.)
                    break
            if not is_custom:
                if name == 'bias' and not (is_norm or is_dcn_module):
                    param_group['lr'] = base_lr * bias_lr_mult
                if prefix.find('conv_offset') != -1 and is_dcn_module and isinstance(module, torch.nn.Conv2d):
                    param_group['lr'] = base_lr * dcn_offset_lr_mult
                if base_wd is not None:
                    if is_norm:
                        param_group['weight_decay'] = base_wd * norm_decay_mult
                    elif is_dwconv:
                        param_group['weight_decay'] = base_wd * dwconv_decay_mult
                    elif name == 'bias' and not is_dcn_module:
                        param_group['weight_decay'] = base_wd * bias_decay_mult
            parameters.append(param_group)
        for child_name, child_mod in module.named_children():
            child_prefix = f'{prefix}.{child_name}' if prefix else child_name
            Constructor.add_params(parameters, child_mod, optimizer_cfg, paramwise_cfg, prefix=child_prefix,
                                   is_dcn_module=False)

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
