"""SegmentationHead (reference ``torchok/models/heads/segmentation/base.py:12-42``): 1x1 classifier conv (bias)
-> bilinear upsample to the input size (``F.interpolate(..., mode='bilinear')``, align_corners=None == False)
-> squeeze for one class."""
from typing import List

import torch
import torch.nn as nn
from torch import Tensor

from ... import engine
from ...constructor import HEADS
from ...engine import functional as EF
from ...engine import resample as ER
from ...losses import cross_entropy as CE
from ..base import BaseModel


@HEADS.register_class
class SegmentationHead(BaseModel):
    def __init__(self, in_channels: int, num_classes: int, do_interpolate: bool = True):
        super().__init__(in_channels, num_classes)
        self.num_classes = num_classes
        self.do_interpolate = do_interpolate
        self.classifier = nn.Conv2d(in_channels, num_classes, kernel_size=1)
        self.init_weights()

    def forward(self, x: List[Tensor]) -> Tensor:
        input_image, features = x
        size = tuple(input_image.shape[2:])
        # training, more than one class: the interpolation is handed to whoever consumes the logits — CrossEntropyLoss fuses
        # it into its kernels (losses/cross_entropy.py: UpsampledLogits), any other consumer materialises it on first touch
        lazy = (self.do_interpolate and self.num_classes > 1 and self.training and torch.is_grad_enabled()
                and CE.FUSE_UPSAMPLE_CE)
        with engine.region() as r:
            logits = EF.conv_bn_act(r, r.input(features), self.classifier, None, False, None)
            if self.do_interpolate and not lazy:
                logits = ER.bilinear_resize(r, logits, size)
            segm_logits = r.output(logits)
        if lazy:
            return CE.UpsampledLogits(segm_logits, size)
        if self.num_classes == 1:
            segm_logits = segm_logits[:, 0]
        return segm_logits
