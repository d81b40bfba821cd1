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
        self.backbone = BACKBONES.get(backbone_name)(**(backbone_params or {}))
        width = self.backbone.out_channels            # channel count handed down the chain
        # a missing stage is an Identity and (reference quirk, classification.py:52-71) the head then takes the
        # BACKBONE width, whatever the neck produced
        self.neck = nn.Identity()
        if neck_name is not None:
            self.neck = NECKS.get(neck_name)(in_channels=self.backbone.out_encoder_channels, **(neck_params or {}))
            width = self.neck.out_channels
        self.pooling = nn.Identity()
        head_width = self.backbone.out_channels
        if pooling_name is not None:
            self.pooling = POOLINGS.get(pooling_name)(in_channels=width, **(pooling_params or {}))
            head_width = self.pooling.out_channels
        self.head = nn.Identity() if head_name is None else \
            HEADS.get(head_name)(in_channels=head_width, **(head_params or {}))

    def _stages(self):
        return self.backbone, self.neck, self.pooling

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for stage in self._stages():
            x = stage(x)
        return self.head(x)

    def forward_with_gt(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Tensor]:
        target = batch.get('target')
        x = batch.get('image')
        for stage in self._stages():
            x = stage(x)
        output = {'embeddings': x, 'prediction': self.head(x, target)}
        if target is not None:
            output['target'] = target
        return output

    def as_module(self) -> nn.Sequential:
        return nn.Sequential(*self._stages(), self.head)
