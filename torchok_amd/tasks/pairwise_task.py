"""PairwiseLearnTask (reference ``torchok/tasks/pairwise_task.py:13-107``): metric learning on
pairwise losses — ``forward_with_gt`` returns ``emb1``, ``emb2`` (the same tensor), the relevance
matrix ``R`` and ``target``.

Deviation (documented): the reference forwards its constructor arguments to
``ClassificationTask.__init__`` POSITIONALLY in the wrong order (``pairwise_task.py:51-52`` vs
``classification.py:16-28``), which makes ``pairwise_sop.yaml`` raise ``KeyError`` in NECKS.  The
documented intent (keyword-correct wiring) is implemented here."""
from typing import Dict

import torch
from torch import Tensor

from .. import _C
from ..constructor import TASKS
from ..engine.core import ptr, require_device, stream_ptr
from .classification import ClassificationTask


@TASKS.register_class
class PairwiseLearnTask(ClassificationTask):
    def __init__(self, hparams, backbone_name: str, pooling_name: str, head_name: str, num_classes: int = None,
                 neck_name: str = None, backbone_params: dict = None, neck_params: dict = None,
                 pooling_params: dict = None, head_params: dict = None, inputs: dict = None):
        super().__init__(hparams, backbone_name=backbone_name, neck_name=neck_name, pooling_name=pooling_name,
                         head_name=head_name, backbone_params=backbone_params, neck_params=neck_params,
                         pooling_params=pooling_params, head_params=head_params, inputs=inputs)
        self.num_classes = num_classes

    def forward_with_gt(self, batch: Dict[str, torch.Tensor]) -> Dict[str, Tensor]:
        input_data = batch.get('image')
        target = batch.get('target')
        embedding = self.forward(input_data)
        output = {'emb1': embedding, 'emb2': embedding}
        if target is not None:
            output['R'] = self.calc_relevance_matrix(target)
            output['target'] = target
        return output

    def calc_relevance_matrix(self, y: Tensor) -> Tensor:
        """R[i, j] = 1.0 where samples i and j share a class.  1-D labels: exact label equality (what
        one-hot scatter -> y y^T -> > 0 evaluates to, :98-105); 2-D multi-label matrices: intersection > 0."""
        require_device(y)
        if y.ndim == 1:
            lab = y.to(torch.int64).contiguous()
            n = lab.shape[0]
            R = torch.empty((n, n), dtype=torch.float32, device=y.device)
            _C.check(_C.lib().tok_relevance_matrix(ptr(lab), ptr(lab), n, n, ptr(R), stream_ptr()),
                     'tok_relevance_matrix')
            return R
        if y.ndim != 2:
            raise ValueError(f'calc_relevance_matrix: labels (N,) or a multi-label matrix (N, L), got {tuple(y.shape)}')
        lab = y.to(torch.float32).contiguous()
        n, classes = lab.shape
        R = torch.empty((n, n), dtype=torch.float32, device=y.device)
        _C.check(_C.lib().tok_relevance_matrix_multilabel(ptr(lab), ptr(lab), n, n, classes, ptr(R), stream_ptr()),
                 'tok_relevance_matrix_multilabel')
        return R
