"""ConvBnAct brick (reference ``torchok/models/modules/bricks/convbnact.py:9-62``): parameter container with
the reference's child names (``conv``, ``bn``, ``act``); executed as ONE fused engine unit."""
from typing import Optional

import torch.nn as nn

from ...engine import functional as EF


class ConvBnAct(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, padding: int = 0, stride: int = 1,
                 bias: bool = False, use_batchnorm: bool = True, groups: int = 1,
                 act_layer: Optional[nn.Module] = nn.ReLU):
        super().__init__()
        if act_layer not in (None, nn.ReLU):
            raise NotImplementedError('torchok_amd ConvBnAct: ReLU or no activation')
        if act_layer is not None and not use_batchnorm:
            raise NotImplementedError('torchok_amd ConvBnAct: activation without BatchNorm')
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels) if use_batchnorm else nn.Identity()
        self.act = act_layer(inplace=True) if act_layer is not None else nn.Identity()

    def run(self, region, x):
        bn = self.bn if isinstance(self.bn, nn.BatchNorm2d) else None
        return EF.conv_bn_act(region, x, self.conv, bn, relu=isinstance(self.act, nn.ReLU))
