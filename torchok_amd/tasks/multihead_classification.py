"""MultiHeadClassificationTask: one backbone + pooling feeding several named heads, each with its own target and an
optional per-sample condition (reference ``torchok/tasks/multihead_classification.py:12-149``; same constructor
arguments and the same ``forward`` / ``forward_with_gt`` outputs: a namedtuple of head outputs, and ``embeddings`` +
``prediction_<head>`` + ``target_<target>``).  As in the reference the neck is constructed (its width feeds the
pooling) but neither forward path calls it (:99-100, :131-132)."""
from collections import namedtuple
from typing import Any, Dict, List

import torch
import torch.nn as nn

from ..constructor import BACKBONES, HEADS, NECKS, POOLINGS, TASKS
from .base import BaseTask


@TASKS.register_class
class MultiHeadClassificationTask(BaseTask):
    def __init__(self, hparams, backbone_name: str, heads: List[Dict[str, Any]], neck_name: str = None,
                 pooling_name: str = None, backbone_params: dict = None, neck_params: dict = None,
                 pooling_params: dict = None, inputs: dict = None):
        super().__init__(hparams, inputs)
        self.backbone = BACKBONES.get(backbone_name)(**(backbone_params or {}))
        pooled_from = self.backbone.out_channels
        self.neck = nn.Identity()
        if neck_name is not None:
            self.neck = NECKS.get(neck_name)(in_channels=self.backbone.out_encoder_channels, **(neck_params or {}))
            pooled_from = self.neck.out_channels
        self.pooling = nn.Identity()
        head_width = self.backbone.out_channels
        if pooling_name is not None:
            self.pooling = POOLINGS.get(pooling_name)(in_channels=pooled_from, **(pooling_params or {}))
            head_width = self.pooling.out_channels
        self.heads = nn.ModuleDict()
        self.target_mapping = {}
        for spec in heads:
            self.heads[spec['name']] = HEADS.get(spec['type'])(in_channels=head_width, **spec['params'])
            self.target_mapping[spec['name']] = spec['target']
        self.head_tuple = namedtuple('HeadOutput', list(self.target_mapping))

    def _embed(self, image: torch.Tensor) -> torch.Tensor:
        return self.pooling(self.backbone(image))

    def forward(self, x: torch.Tensor):
        features = self._embed(x)
        return self.head_tuple(**{name: head(features) for name, head in self.heads.items()})

    def forward_with_gt(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        features = self._embed(batch['image'])
        output = {'embeddings': features}
        for name, head in self.heads.items():
            target_name = self.target_mapping[name]
            target = batch[f'target_{target_name}']
            condition = batch.get(f'condition_{target_name}')
            rows = features
            if condition is not None:       # only the samples that carry this label reach the head (:137-139)
                target = target[condition]
                rows = features[condition]
            output[f'prediction_{name}'] = head(rows, target)
            output[f'target_{target_name}'] = target
        return output

    def as_module(self) -> nn.Sequential:
        raise NotImplementedError()
