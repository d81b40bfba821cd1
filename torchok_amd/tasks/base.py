"""BaseTask: the Lightning-module surface of the reference (``torchok/tasks/base.py:17-204``)
without the lightning dependency (absent here).  Same per-step API — ``training_step`` /
``validation_step`` / ``test_step`` / ``predict_step`` / ``configure_optimizers`` /
``forward_with_gt`` / ``as_module`` — same return dicts, same example-input buffers
(``input_tensors_{i}``, :37-43).  ``self.log`` is a cheap recorder (no host sync)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Union

import torch
import torch.nn as nn
from torch import Tensor

from ..constructor.config import Phase
from ..constructor.constructor import Constructor


class BaseTask(nn.Module, ABC):
    def __init__(self, hparams, inputs=None, **kwargs):
        super().__init__()
        self._hparams = hparams
        self._constructor = Constructor(hparams)
        self.input_tensor_names = []
        self.losses = self._constructor.configure_losses() if hparams.get('joint_loss') is not None else None
        self.metrics_manager = self._constructor.configure_metrics_manager()
        self.example_input_array = []
        self.logged: Dict[str, Tensor] = {}
        self.current_epoch = 0
        if inputs is not None:
            for i, input_params in enumerate(inputs):
                name = f'input_tensors_{i}'
                self.input_tensor_names.append(name)
                t = torch.rand(1, *input_params['shape']).type(torch.__dict__[input_params['dtype']])
                self.example_input_array.append(t)
                self.register_buffer(name, t)

    @property
    def hparams(self):
        return self._hparams

    def log(self, name: str, value, **kwargs) -> None:
        self.logged[name] = value.detach() if isinstance(value, Tensor) else value

    def log_dict(self, d: Dict[str, Any], **kwargs) -> None:
        for k, v in d.items():
            self.log(k, v)

    @abstractmethod
    def forward(self, *args, **kwargs) -> torch.Tensor:
        pass

    @abstractmethod
    def forward_with_gt(self, batch: Dict[str, Any]) -> Dict[str, torch.Tensor]:
        pass

    @abstractmethod
    def as_module(self) -> nn.Sequential:
        pass

    def configure_optimizers(self) -> List[Dict[str, Any]]:
        return self._constructor.configure_optimizers(list(self.children()))

    def training_step(self, batch: Dict[str, Union[Tensor, int]], batch_idx: int) -> Dict[str, Tensor]:
        output = self.forward_with_gt(batch)
        total_loss, tagged_loss_values = self.losses(**output)
        self.log('loss', total_loss, prog_bar=True, on_step=True)
        self.metrics_manager.update(Phase.TRAIN, **output)
        output_dict = {'loss': total_loss}
        output_dict.update(tagged_loss_values)
        return output_dict

    def validation_step(self, batch, batch_idx: int, dataloader_idx: int = 0) -> Dict[str, Tensor]:
        output = self.forward_with_gt(batch)
        self.metrics_manager.update(Phase.VALID, dataloader_idx, **output)
        if self._hparams.task.compute_loss_on_valid:
            total_loss, tagged_loss_values = self.losses(**output)
            output_dict = {'loss': total_loss}
            self.log('loss', total_loss, prog_bar=True, on_step=True)
            output_dict.update(tagged_loss_values)
        else:
            output_dict = {}
        return output_dict

    def test_step(self, batch, batch_idx: int, dataloader_idx: int = 0) -> None:
        output = self.forward_with_gt(batch)
        self.metrics_manager.update(Phase.TEST, dataloader_idx, **output)

    def predict_step(self, batch, batch_idx: int, dataloader_idx: int = 0) -> Dict[str, Tensor]:
        return self.forward_with_gt(batch)

    # ---- per-batch / per-epoch hooks (reference tasks/base.py:163-200) -----------------------------------------
    def _mean_over_ranks_async(self, outputs: Dict[str, Tensor]):
        """The reference all_gathers every logged value and takes the mean (tasks/base.py:170,182).  Here the values are
        stacked once and mean-all-reduced as ONE tiny collective; on a GPU job with a GradientAllReducer attached
        (`strategy: ddp`) that collective goes onto the reducer's comm stream and is consumed a step later — no host
        synchronisation and nothing blocking on the compute stream.  Returns (tags, values, work or None, scale)."""
        import torch.distributed as dist
        tags = list(outputs)
        if not tags:
            return tags, None, None, 1.0
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            # one rank: the mean over ranks is the value itself; a scalar costs no kernel, anything else is reduced to the
            # scalar the multi-rank path (and the reference's `.mean()`, tasks/base.py:170) would log
            return tags, [v if v.dim() == 0 else v.float().mean() for v in (outputs[t].detach() for t in tags)], None, 1.0
        vals = torch.stack([outputs[t].detach().float().mean() for t in tags])
        red = getattr(self, '_grad_reducer', None)
        if red is not None:
            work, vals, scale = red.mean_small_async(vals)
            return tags, vals, work, scale
        dist.all_reduce(vals)
        return tags, vals, None, 1.0 / dist.get_world_size()

    def _mean_over_ranks(self, outputs: Dict[str, Tensor]) -> Dict[str, Tensor]:
        tags, vals, work, scale = self._mean_over_ranks_async(outputs)
        if not tags:
            return {}
        if work is not None:
            work.wait()
        if scale != 1.0:
            vals = vals * scale
        return {t: vals[i] for i, t in enumerate(tags)}

    def flush_step_logs(self) -> Dict[str, Tensor]:
        """Consume the pending per-step loss mean (issued by the previous on_train_batch_end): the current stream waits for
        the collective, the values are logged under `train/<tag>`.  Called by the next on_train_batch_end and at epoch end."""
        pend = getattr(self, '_pending_loss_mean', None)
        if pend is None:
            return {}
        self._pending_loss_mean = None
        tags, vals, work, scale, n = pend
        if work is not None:
            work.wait()
        if scale != 1.0:
            vals = vals * scale
        out = {t: vals[i] for i, t in enumerate(tags)}
        for tag, value in out.items():
            self.log(f'train/{tag}', value, on_step=False, on_epoch=True, batch_size=n)
        return out

    def on_train_batch_end(self, outputs: Dict[str, Tensor], batch, batch_idx: int, dataloader_idx: int = 0):
        """reference tasks/base.py:163-173.  The collective of THIS step is issued now and consumed at the next call (or at
        epoch end): returned / logged values lag one step on a multi-rank job and are immediate on one rank."""
        prev = self.flush_step_logs()
        tags, vals, work, scale = self._mean_over_ranks_async(outputs)
        if not tags:
            return prev
        self._pending_loss_mean = (tags, vals, work, scale, len(outputs))
        if work is None:
            return self.flush_step_logs()
        return prev

    def on_validation_batch_end(self, outputs: Dict[str, Tensor], batch, batch_idx: int, dataloader_idx: int = 0):
        output_dict = self._mean_over_ranks(outputs)
        for tag, value in output_dict.items():
            self.log(f'valid/{tag}', value, on_step=False, on_epoch=True, batch_size=len(outputs))
        return output_dict

    def on_train_epoch_end(self) -> None:
        self.flush_step_logs()
        self.log_dict(self.metrics_manager.on_epoch_end(Phase.TRAIN))
        self.log('step', float(self.current_epoch), on_step=False, on_epoch=True)

    def on_validation_epoch_end(self) -> None:
        self.log_dict(self.metrics_manager.on_epoch_end(Phase.VALID))
        self.log('step', float(self.current_epoch), on_step=False, on_epoch=True)

    def on_test_epoch_end(self) -> None:
        self.log_dict(self.metrics_manager.on_epoch_end(Phase.TEST))
