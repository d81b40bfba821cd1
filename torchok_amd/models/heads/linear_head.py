"""Linear / classification heads (reference ``torchok/models/heads/representation/linear_head.py:10-36``,
``heads/classification/classification_head.py:9-40``)."""
from typing import Optional

import torch
import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import HEADS
from ...engine import functional as EF
from ...engine import metric as EM
from ..base import BaseModel


@HEADS.register_class
class LinearHead(BaseModel):
    def __init__(self, in_channels, out_channels, drop_rate=0.0, bias=True, normalize=False):
        super().__init__(in_channels, out_channels)
        self.drop_rate = drop_rate
        self.normalize = normalize
        self.fc = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x: Tensor, targets: Optional[Tensor] = None) -> Tensor:
        if self.drop_rate > 0. and self.training:
            raise NotImplementedError('torchok_amd LinearHead: dropout is not built (p = 0 in all hot-path configs)')
        with engine.region() as r:
            y = EF.linear(r, r.input(x), self.fc)
            if self.normalize:
                y = EM.l2_normalize(r, y)
            return r.output(y)


@HEADS.register_class
class ClassificationHead(LinearHead):
    def __init__(self, in_channels: int, num_classes: int, drop_rate: float = 0.0, bias: bool = True):
        super().__init__(in_channels, out_channels=num_classes, drop_rate=drop_rate, bias=bias)

    def forward(self, x: Tensor, target: Optional[Tensor] = None) -> Tensor:
        x = super().forward(x, target)
        if self.out_channels == 1:
            x = x[..., 0]
        return x
