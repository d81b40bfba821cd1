"""Base classes of TorchOk models (reference ``torchok/models/base.py:8-63`` and
``torchok/models/backbones/base_backbone.py:11-64``): the in/out-channel contract that Tasks use
to wire backbone -> neck -> pooling -> head."""
from abc import ABC, abstractmethod
from typing import List, Optional, Tuple, Union

import torch.nn as nn
from torch import Tensor


class BaseModel(nn.Module, ABC):
    def __init__(self, in_channels=None, out_channels=None):
        super().__init__()
        self._in_channels = in_channels
        self._out_channels = out_channels

    @abstractmethod
    def forward(self, *args, **kwargs) -> Tensor:
        pass

    def no_weight_decay(self) -> List[str]:
        return list()

    @property
    def in_channels(self):
        if self._in_channels is None:
            raise ValueError('TorchOk Models must have self._in_channels attribute.')
        return self._in_channels

    @property
    def out_channels(self):
        if self._out_channels is None:
            raise ValueError('TorchOk Models must have self._out_channels attribute.')
        return self._out_channels

    def init_weights(self):
        # reference models/base.py:50-63
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode='fan_in', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


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

    @property
    def out_encoder_channels(self) -> Tuple[int]:
        if self._out_encoder_channels is None:
            raise ValueError('TorchOk Backbones must have self._out_feature_channels attribute.')
        return tuple(self._out_encoder_channels)

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
