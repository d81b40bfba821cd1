"""HRNetSegmentationNeck (reference ``torchok/models/necks/segmentation/hrnet.py:16-43``): bilinear
(align_corners=False) upsample of the three low-resolution branches to branch-0 size, channel concat,
1x1 ConvBnReLU.  Here as ONE engine unit in the commuted order (engine/neck.py): the pointwise product runs at every
source's own resolution and the four results are interpolated and summed — the concat tensor is never built.  Where
that form is not served (BatchNorm in eval mode, widths that are not multiples of 8; TOK_NECK_COMMUTE=0) the four sources are
interpolated straight into their channel slices of ONE buffer, then one fused conv-BN-ReLU unit."""
from functools import partial
from typing import List, Tuple, Union

import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import NECKS
from ...engine import neck as EN
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
            cba = self.convbnact
            bn = cba.bn if isinstance(cba.bn, nn.BatchNorm2d) else None
            feats = r.output(EN.upsample_concat_conv_bn_relu(r, srcs, (x0.size(2), x0.size(3)), cba.conv, bn,
                                                             relu=isinstance(cba.act, nn.ReLU)))
        return [input_image, feats]
