"""HRNetSegmentationNeck (reference ``torchok/models/necks/segmentation/hrnet.py:16-43``): bilinear
(align_corners=False) upsample of the three low-resolution branches to branch-0 size, channel concat,
1x1 ConvBnReLU.  Here the four sources are interpolated straight into their channel slices of ONE buffer
(no separate upsampled maps, no torch.cat pass), then one fused conv-BN-ReLU unit."""
from functools import partial
from typing import List, Tuple, Union

import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import NECKS
from ...engine import resample as ER
from ..base import BaseModel
from ..modules import ConvBnAct

ConvBnRelu = partial(ConvBnAct, act_layer=nn.ReLU)


@NECKS.register_class
class HRNetSegmentationNeck(BaseModel):
    def __init__(self, in_channels: Union[List[int], Tuple[int, ...]]):
        out_channels = sum(in_channels)
        super().__init__(in_channels, out_channels)
        self.convbnact = ConvBnRelu(out_channels, out_channels, kernel_size=1, padding=0, stride=1)

    def forward(self, features: List[Tensor]) -> List[Tensor]:
        input_image, x0, x1, x2, x3 = features
        with engine.region() as r:
            srcs = [r.input(t) for t in (x0, x1, x2, x3)]
            feats = ER.bilinear_concat(r, srcs, (x0.size(2), x0.size(3)))
            feats = r.output(self.convbnact.run(r, feats))
        return [input_image, feats]
