"""hipGraph capture of a whole training step.

A step of a deep multi-branch network (HRNet-W48: ~4000 kernel launches) is bound by the host's launch rate,
not by the GPU.  `GraphedTrainingStep` records ONE step — `task.training_step` -> backward -> gradient exchange
-> `optimizer.step()` — into a hipGraph on static input buffers and replays it: one host call per step, kernels
back to back on the device.  (This replaces nothing in the reference, which relies on eager PyTorch; it is the
MI355X-native answer to its per-op launch overhead — streams and graphs instead of a tracing compiler.)

Constraints (checked where possible): static shapes; no host synchronisation inside `training_step`; scalar
hyper-parameters are baked into the recording, so call `recapture()` after changing the learning rate; SGD, or
Adam / AdamW built with capturable=True (device-side step count, tok_adam_step_capturable)."""
from typing import Dict, Optional

import torch


class GraphedTrainingStep:
    def __init__(self, task, optimizer, example_batch: Dict[str, torch.Tensor], reducer=None, warmup: int = 3):
        from ..optim.optimizers import SGD, _AdamBase
        capt = isinstance(optimizer, _AdamBase) and all(g.get('capturable', False) for g in optimizer.param_groups)
        if not isinstance(optimizer, SGD) and not capt:
            raise NotImplementedError('GraphedTrainingStep: fused SGD, or Adam / AdamW with capturable=True (the plain Adam step '
                                      'takes its step count as a host scalar, which a recording would freeze)')
        self.task, self.optimizer, self.reducer = task, optimizer, reducer
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.out: Optional[Dict[str, torch.Tensor]] = None
        self._warmup = warmup
        self._step_idx = 0
        self.recapture()

    def _eager(self):
        from .step import train_step
        # (on_train_batch_end's bookkeeping is host-side state: it runs outside the recording, see __call__)
        out = train_step(self.task, self.optimizer, self.static, self._step_idx, self.reducer, batch_end_hook=False)
        self._step_idx += 1
        return out

    def recapture(self):
        """(Re)record the step: a few eager steps on a side stream first (arena construction, momentum
        initialisation, pack tables, allocator warm-up), then the capture."""
        from . import core
        core.note_main_stream()     # the streams the engine probes are picked relative to the caller's stream, not the warm-up's
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._eager()

    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for k, v in batch.items():
            if torch.is_tensor(v):
                if self.static[k].shape != v.shape or self.static[k].dtype != v.dtype:
                    raise ValueError(f'GraphedTrainingStep: batch["{k}"] changed shape/dtype; static shapes only')
                if v.data_ptr() != self.static[k].data_ptr():
                    self.static[k].copy_(v, non_blocking=True)
        self.graph.replay()
        self._step_idx += 1
        self.task.on_train_batch_end(self.out, batch, self._step_idx - 1)
        return self.out
