"""ArcFace margin head (reference ``torchok/models/heads/classification/arcface_head.py:20-131``).

train:  normalize(x) . normalize(W)^T -> margin on the target column -> x scale   (:110-131, :95-108)
eval :  plain ``F.linear(input, weight)`` — no normalisation (:120-121)
defaults: scale = (C-1)/C * ln((C-1) p/(1-p)) + 1 with p = .999 (:47-50); margin = .5 C/(C-1), or
.9 - cos(2 pi / C) for 2-D embeddings (:52-56).  ``dynamic_margin=True`` cannot run in the reference
either (buffer registered as ``step`` but read as the name-mangled ``__step``, :69 vs :85,93 —
SURVEY.md App. B.4): it is rejected here at the same point (first training forward)."""
import math
from typing import Optional

import torch
import torch.nn as nn

from ... import engine
from ...constructor import HEADS
from ...engine import functional as EF
from ...engine import metric as EM
from ..base import BaseModel


@HEADS.register_class
class ArcFaceHead(BaseModel):
    def __init__(self, in_channels: int, num_classes: int, scale: float = None, margin: float = None,
                 easy_margin: bool = False, dynamic_margin: bool = False, num_warmup_steps: int = None,
                 min_margin: float = None):
        super().__init__(in_channels, out_channels=num_classes)
        if scale is None:
            p = .999
            c_1 = (num_classes - 1)
            scale = c_1 / num_classes * math.log(c_1 * p / (1 - p)) + 1
        if margin is None:
            if in_channels == 2:
                margin = .9 - math.cos(2 * math.pi / num_classes)
            else:
                margin = .5 * num_classes / (num_classes - 1)
        self.dynamic_margin = dynamic_margin
        if self.dynamic_margin:
            if num_warmup_steps is None or not isinstance(num_warmup_steps, int):
                raise ValueError('`num_warmup_steps` must be positive int when `dynamic_margin` is True')
            if min_margin is None:
                raise ValueError('`min_margin` must be float when `dynamic_margin` is True')
            self.num_warmup_steps = num_warmup_steps
            self.min_margin = min_margin
            self.max_margin = margin
            self.margin = min_margin
            self.register_buffer('step', torch.tensor(0))
        else:
            self.margin = margin
        self.scale = scale
        self.easy_margin = easy_margin
        self.weight = nn.Parameter(torch.zeros(num_classes, in_channels), requires_grad=True)
        nn.init.xavier_uniform_(self.weight)
        # eval path = a bias-free Linear over the same parameter (not registered twice)
        lin = nn.Linear(in_channels, num_classes, bias=False)
        lin.weight = self.weight
        object.__setattr__(self, '_eval_linear', lin)

    def forward(self, input: torch.Tensor, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not self.training:
            self._eval_linear.weight = self.weight
            with engine.region() as r:
                return r.output(EF.linear(r, r.input(input), self._eval_linear))
        elif target is None:
            raise ValueError('Target is None in training mode.')
        if self.dynamic_margin:
            raise AttributeError("'ArcFaceHead' object has no attribute '_ArcFaceHead__step' "
                                 '(dynamic_margin is broken in the reference as well; use a static margin)')
        with engine.region() as r:
            x = EM.l2_normalize(r, r.input(input))
            cosine = EM.cosine_linear(r, x, self.weight)
            return r.output(EM.arcface_margin(r, cosine, target, self.margin, self.scale, self.easy_margin))
