"""SegmentationTask (reference ``torchok/tasks/segmentation.py:12-98``): backbone.forward_features -> neck ->
head; ``forward_with_gt`` returns ``prediction`` (+ ``target``)."""
from typing import Dict

import torch
import torch.nn as nn

from ..constructor import BACKBONES, HEADS, NECKS, TASKS
from ..models.base import BackboneWrapper
from .base import BaseTask


@TASKS.register_class
class SegmentationTask(BaseTask):
    def __init__(self, hparams, backbone_name: str, head_name: str, neck_name: str, backbone_params: dict = None,
                 neck_params: dict = None, head_params: dict = None, **kwargs):
        super().__init__(hparams, **kwargs)
        self.backbone = BACKBONES.get(backbone_name)(**(backbone_params or dict()))
        self.neck = NECKS.get(neck_name)(in_channels=self.backbone.out_encoder_channels, **(neck_params or dict()))
        self.head = HEADS.get(head_name)(in_channels=self.neck.out_channels, **(head_params or dict()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.backbone.forward_features(x)
        x = self.neck(x)
        return self.head(x)

    def forward_with_gt(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        input_data = batch.get('image')
        target = batch.get('target')
        features = self.backbone.forward_features(input_data)
        neck_out = self.neck(features)
        prediction = self.head(neck_out)
        output = {'prediction': prediction}
        if target is not None:
            output['target'] = target
        return output

    def as_module(self) -> nn.Sequential:
        return nn.Sequential(BackboneWrapper(self.backbone), self.neck, self.head)
