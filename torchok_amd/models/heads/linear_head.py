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
from ...engine import ocr as EO
from ..base import BaseModel


@HEADS.register_class
class LinearHead(BaseModel):
    def __init__(self, in_channels, out_channels, drop_rate=0.0, bias=True, normalize=False):
        super().__init__(in_channels, out_channels)
        self.drop_rate = drop_rate
        self.normalize = normalize
        self.fc = nn.Linear(in_channels, out_channels, bias=bias)

    def draw_dropout(self, rows: int, device) -> Optional[Tensor]:
        """Keep/scale factors of F.dropout on the (rows, in_channels) embedding: one Bernoulli(1 - p) draw per element,
        kept elements scaled by 1 / (1 - p); None outside training or at p = 0 (linear_head.py:27-28)."""
        if not (self.drop_rate > 0. and self.training):
            return None
        if self.drop_rate >= 1.:
            return torch.zeros((rows, self.in_channels), dtype=torch.float32, device=device)
        keep = 1. - self.drop_rate
        return torch.empty((rows, self.in_channels), dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x: Tensor, targets: Optional[Tensor] = None) -> Tensor:
        with engine.region() as r:
            xin = r.input(x)
            # an embedding row is an image of one pixel: element dropout = the per-(image, channel) scale kernel
            xin = EO.channel_dropout(r, xin, self.draw_dropout(xin.shape[0], xin.data.device))
            y = EF.linear(r, xin, self.fc)
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
