"""HRNetClassificationNeck (reference ``torchok/models/necks/classification/hrnet.py:12-92``).

Faithful to the reference forward (:83-92), including its quirk: inside the loop ``y`` is OVERWRITTEN by
``incre_modules[i + 1](x[i + 1])`` (the upstream HRNet adds), so the result is
``final_layer(incre_modules[3](x[3]))``; the down-sampling modules still run (their BatchNorm running statistics
move in training, exactly as in the reference) but feed nothing and receive no gradient."""
from typing import List, Tuple, Union

import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import NECKS
from ..backbones.resnet import Bottleneck
from ..base import BaseModel
from ..modules import ConvBnAct


@NECKS.register_class
class HRNetClassificationNeck(BaseModel):
    def __init__(self, in_channels: Union[List[int], Tuple[int, ...]]):
        super().__init__(in_channels, 2048)
        self.head_channels = [32, 64, 128, 256]
        exp = Bottleneck.expansion
        self.incre_modules = nn.ModuleList(self._make_layer(c, self.head_channels[i]) for i, c in enumerate(in_channels))
        self.downsamp_modules = nn.ModuleList(
            ConvBnAct(in_channels=self.head_channels[i] * exp, out_channels=self.head_channels[i + 1] * exp,
                      kernel_size=3, padding=1, stride=2) for i in range(len(in_channels) - 1))
        self.final_layer = ConvBnAct(in_channels=self.head_channels[3] * exp, out_channels=self.out_channels,
                                     kernel_size=1, padding=0, stride=1)

    @staticmethod
    def _make_layer(inplanes: int, planes: int) -> nn.Sequential:
        downsample = None
        if inplanes != planes * Bottleneck.expansion:
            downsample = ConvBnAct(in_channels=inplanes, out_channels=planes * Bottleneck.expansion, kernel_size=1,
                                   padding=0, stride=1, bias=False, act_layer=None)
        return nn.Sequential(Bottleneck(inplanes, planes, 1, downsample))

    def forward(self, x: List[Tensor]) -> Tensor:
        with engine.region() as r:
            xs = [r.input(t) for t in x]
            y = self.incre_modules[0](xs[0])
            for i, down in enumerate(self.downsamp_modules):
                y = down.run(r, y)
                if i + 1 < len(xs):
                    y = self.incre_modules[i + 1](xs[i + 1])
            return r.output(self.final_layer.run(r, y))
