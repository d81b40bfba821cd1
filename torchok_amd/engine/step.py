"""ONE training step, as Lightning's fit loop drives a reference Task:

    training_step (tasks/base.py:125-133) -> zero_grad -> backward (bucketed gradient exchange overlapped with it)
    -> optimizer.step -> on_train_batch_end (tasks/base.py:163-173: the per-step loss mean over ranks)

`bench.py`, `torchok_amd.run.fit`, `GraphedTrainingStep` and the world-2 tests all call THIS function, so the code path
the driver times on 8 GPUs is the one the CPU tests execute."""
from typing import Dict, Optional

import torch


def train_step(task, optimizer, batch: Dict[str, torch.Tensor], batch_idx: int, reducer=None,
               batch_end_hook: bool = True) -> Dict[str, torch.Tensor]:
    out = task.training_step(batch, batch_idx)
    optimizer.zero_grad(set_to_none=True)
    if reducer is not None:
        reducer.begin_step()
    out['loss'].backward()
    if reducer is not None:
        reducer.finish_step()
    optimizer.step()
    if batch_end_hook:
        task.on_train_batch_end(out, batch, batch_idx)
    return out


def replicas_in_sync(reducer) -> Optional[bool]:
    """True iff every rank holds bit-identical parameters (one tiny MIN/MAX collective over a float64 checksum); None
    without a reducer.  Host read: call it outside timed regions."""
    if reducer is None:
        return None
    lo_hi = reducer.params_checksum()
    return bool(lo_hi[0].item() == lo_hi[1].item())
