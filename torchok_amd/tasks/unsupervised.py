"""SimCLRTask (reference ``torchok/tasks/simclr_task.py:9-82``) and TripletLearnTask (``tasks/triplet_task.py:11-50``):
ClassificationTask wiring with two / three forwards per step (weight gradients accumulate in the parameter slots)."""
from typing import Dict, Union

from torch import Tensor

from ..constructor import TASKS
from .classification import ClassificationTask


@TASKS.register_class
class SimCLRTask(ClassificationTask):
    def __init__(self, hparams, backbone_name: str, pooling_name: str = None, head_name: str = None, neck_name: str = None,
                 backbone_params: dict = None, neck_params: dict = None, pooling_params: dict = None,
                 head_params: dict = None, inputs: dict = None):
        # (the reference forwards these positionally in an order ClassificationTask does not have, simclr_task.py:51-62;
        #  the documented wiring is keyword-correct here — same deviation as PairwiseLearnTask)
        super().__init__(hparams, backbone_name=backbone_name, neck_name=neck_name, pooling_name=pooling_name,
                         head_name=head_name, backbone_params=backbone_params, neck_params=neck_params,
                         pooling_params=pooling_params, head_params=head_params, inputs=inputs)

    def forward_with_gt(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        x1, x2 = batch['image_0'], batch['image_1']
        return {'emb1': self.forward(x1), 'emb2': self.forward(x2)}


@TASKS.register_class
class TripletLearnTask(ClassificationTask):
    def __init__(self, hparams, **kwargs):
        super().__init__(hparams, **kwargs)

    def forward_with_gt(self, batch: Dict[str, Union[Tensor, int]]) -> Dict[str, Tensor]:
        return {'anchor': self.forward(batch.get('anchor')), 'positive': self.forward(batch.get('positive')),
                'negative': self.forward(batch.get('negative'))}

    def validation_step(self, batch, batch_idx: int, dataloader_idx: int = 0) -> Dict[str, Tensor]:
        output = ClassificationTask.forward_with_gt(self, batch)
        self.metrics_manager.update('valid', **output)
        if self._hparams.task.compute_loss_on_valid:
            total_loss, tagged = self.losses(**output)
            out = {'loss': total_loss}
            out.update(tagged)
            return out
        return {}
