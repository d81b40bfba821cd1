"""Global pooling (reference ``torchok/models/poolings/classification/pooling.py:7-12`` = [timm]
SelectAdaptivePool2d(flatten=True); ``linear.py:8-25``)."""
import torch
import torch.nn as nn

from ... import engine
from ...constructor import POOLINGS
from ...engine import functional as EF
from ..base import BaseModel


@POOLINGS.register_class
class Pooling(BaseModel):
    def __init__(self, in_channels: int, pooling_type: str = 'avg', output_size: int = 1):
        super().__init__(in_channels, in_channels if pooling_type != 'catavgmax' else 2 * in_channels)
        if pooling_type not in ('avg', 'max', 'avgmax', 'catavgmax'):
            raise ValueError(f'Invalid pool type: {pooling_type}')      # [timm] SelectAdaptivePool2d's assert
        if output_size != 1:
            raise NotImplementedError('torchok_amd Pooling: output_size=1 (global pooling) only')
        self.pool_type = pooling_type

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with engine.region() as r:
            return r.output(EF.global_pool(r, r.input(x), self.pool_type))


@POOLINGS.register_class
class PoolingLinear(Pooling):
    def __init__(self, in_channels, out_channels, pooling_type: str = 'avg', output_size: int = 1, bias=True):
        super().__init__(in_channels, pooling_type, output_size=output_size)
        self.fc = nn.Linear(self._out_channels, out_channels, bias=bias)
        self._out_channels = out_channels
        self.init_weights()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with engine.region() as r:
            t = EF.global_pool(r, r.input(x), self.pool_type)
            return r.output(EF.linear(r, t, self.fc))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
