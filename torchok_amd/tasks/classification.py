"""ClassificationTask: backbone -> neck -> pooling -> head wiring through the registries
(reference ``torchok/tasks/classification.py:13-123``; same constructor arguments, same
``forward`` / ``forward_with_gt`` outputs: ``embeddings``, ``prediction``, ``target``)."""
from typing import Dict

import torch
import torch.nn as nn
from torch import Tensor

from ..constructor import BACKBONES, HEADS, NECKS, POOLINGS, TASKS
from .base import BaseTask


@TASKS.register_class
class ClassificationTask(BaseTask):
    def __init__(self, hparams, backbone_name: str, neck_name: str = None, pooling_name: str = None,
                 head_name: str = None, backbone_params: dict = None, neck_params: dict = None,
                 pooling_params: dict = None, head_params: dict = None, inputs: dict = None):
        super().__init__(hparams, inputs)
        self.backbone = BACKBONES.get(backbone_name)(**(backbone_params or dict()))
        if neck_name is None:
            self.neck = nn.Identity()
            pooling_in_channels = self.backbone.out_channels
        else:
            self.neck = NECKS.get(neck_name)(in_channels=self.backbone.out_encoder_channels, **(neck_params or dict()))
            pooling_in_channels = self.neck.out_channels
        if pooling_name is None:
            self.pooling = nn.Identity()
            head_in_channels = self.backbone.out_channels
        else:
            self.pooling = POOLINGS.get(pooling_name)(in_channels=pooling_in_channels, **(pooling_params or dict()))
            head_in_channels = self.pooling.out_channels
        if head_name is None:
            self.head = nn.Identity()
        else:
            self.head = HEADS.get(head_name)(in_channels=head_in_channels, **(head_params or dict()))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.backbone(x)
        x = self.neck(x)
        x = self.pooling(x)
        x = self.head(x)
        return x

    def forward_with_gt(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Tensor]:
        input_data = batch.get('image')
        target = batch.get('target')
        features = self.backbone(input_data)
        features = self.neck(features)
        embeddings = self.pooling(features)
        prediction = self.head(embeddings, target)
        output = {'embeddings': embeddings, 'prediction': prediction}
        if target is not None:
            output['target'] = target
        return output

    def as_module(self) -> nn.Sequential:
        return nn.Sequential(self.backbone, self.neck, self.pooling, self.head)
