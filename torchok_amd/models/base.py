"""Base classes of TorchOk models (reference ``torchok/models/base.py:8-63`` and
``torchok/models/backbones/base_backbone.py:11-64``): the in/out-channel contract that Tasks use
to wire backbone -> neck -> pooling -> head."""
from abc import ABC, abstractmethod
from typing import List, Optional, Tuple, Union

import torch.nn as nn
from torch import Tensor


def _required(attr: str, message: str, as_tuple: bool = False):
    """Read-only property over a private attribute that must have been set by the subclass constructor."""
    def getter(self):
        value = getattr(self, attr, None)
        if value is None:
            raise ValueError(message)
        return tuple(value) if as_tuple else value
    return property(getter)


class BaseModel(nn.Module, ABC):
    """in_channels / out_channels contract used by the Tasks to chain backbone -> neck -> pooling -> head."""
    in_channels = _required('_in_channels', 'TorchOk Models must have self._in_channels attribute.')
    out_channels = _required('_out_channels', 'TorchOk Models must have self._out_channels attribute.')

    def __init__(self, in_channels=None, out_channels=None):
        super().__init__()
        self._in_channels, self._out_channels = in_channels, out_channels

    @abstractmethod
    def forward(self, *args, **kwargs) -> Tensor:
        ...

    def no_weight_decay(self) -> List[str]:
        return []

    def init_weights(self):
        """Default initialisation of heads / necks (reference models/base.py:50-63): Kaiming-uniform convs,
        unit BatchNorm, Xavier-uniform linears, zero biases."""
        for m in self.modules():
            bias = getattr(m, 'bias', None)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode='fan_in', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
            else:
                continue
            if bias is not None:
                nn.init.constant_(bias, 0)


class BaseBackbone(BaseModel, ABC):
    """Backbone contract: ``forward(x)`` -> last feature map, ``forward_features(x)`` ->
    ``[x] + per-stage features`` (reference: [timm] FeatureHooks on ``feature_info`` modules,
    ``base_backbone.py:14-34``; here the stages are returned by the engine region directly)."""

    feature_info: list

    def create_hooks(self):
        self.stage_names = [h['module'] for h in self.feature_info]
        self._out_encoder_channels = [h['num_chs'] for h in self.feature_info]

    @abstractmethod
    def forward_features(self, x: Tensor) -> List[Tensor]:
        pass

    out_encoder_channels = _required('_out_encoder_channels',
                                     'TorchOk Backbones must have self._out_feature_channels attribute.', as_tuple=True)

    @abstractmethod
    def get_stages(self, stage: int) -> nn.Module:
        pass


class BackboneWrapper(nn.Module):
    def __init__(self, backbone):
        super().__init__()
        self.backbone = backbone

    def forward(self, x):
        return self.backbone.forward_features(x)

    @property
    def out_encoder_channels(self):
        return self.backbone.out_encoder_channels
