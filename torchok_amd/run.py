"""Thin YAML-driven fit loop: what `python -m torchok -cp <dir> -cn <cfg>` does for the training hot path
(reference ``torchok/__main__.py:13-54`` + ``constructor/runner.py:7-19`` + Lightning's fit loop, none of which are on
either box), without the data pipeline (datasets / transforms are out of scope: batches come from the caller or are
synthetic, shaped by ``task.params.inputs`` and the head's class count).

    python -m torchok_amd.run -cp examples/configs -cn classification_imagenet trainer.precision=bf16 trainer.max_steps=10

What of ``trainer`` (config_structure.py:136-171) is CONSUMED here, and how:
  * ``precision``: this build stores activations in bf16 and accumulates in fp32 (the arithmetic of Lightning's
    ``precision: bf16``).  ``bf16`` / ``bf16-mixed`` select it; ``16`` / ``16-mixed`` (fp16 autocast in the reference) run
    the same bf16 path with a logged note; ``32`` / ``64`` — the schema default is 32 — raise ``ValueError``: a drop-in must not
    change the arithmetic a recipe asked for without saying so.
  * ``strategy: ddp`` (+ ``devices: N``): one process per GPU, `GradientAllReducer` (bucketed RCCL all-reduce overlapped
    with backward, rank-0 buffer broadcast, per-step loss mean on the comm stream).  Under torchrun the ranks already
    exist (RANK / WORLD_SIZE); otherwise ``devices: N > 1`` spawns them.  Other strategies raise.
  * ``max_steps`` / ``max_epochs`` / ``limit_train_batches`` bound the loop; ``accumulate_grad_batches != 1``,
    ``gradient_clip_val`` and ``sync_batchnorm: true`` raise (not built); everything else is orchestration and ignored.
``LitTask`` wraps a Task as a ``pytorch_lightning.LightningModule`` iff lightning is importable (it is not here)."""
import argparse
import logging
import os
import sys
from typing import Any, Callable, Dict, Iterable, Optional

import torch

from .constructor import TASKS
from .constructor.config import ConfigDict, load_config
from .engine.step import replicas_in_sync, train_step

log = logging.getLogger('torchok_amd.run')

_BF16 = ('bf16', 'bf16-mixed', 'bf16-true')
_FP16 = ('16', '16-mixed', '16-true')


def resolve_precision(trainer: Dict[str, Any]) -> str:
    """`trainer.precision` -> the arithmetic this build runs, or ValueError (see the module docstring)."""
    p = str(trainer.get('precision', 32))      # schema default (config_structure.py:141)
    if p in _BF16:
        return 'bf16'
    if p in _FP16:
        log.warning('trainer.precision=%s asks for fp16 autocast; this build computes in bf16 storage / fp32 accumulation '
                    '(same exponent range as fp32, no loss scaling needed) and runs the recipe that way', p)
        return 'bf16'
    if p in ('32', '64', '32-true', '64-true'):
        raise ValueError(f'trainer.precision={p}: torchok_amd stores activations in bf16 (fp32 accumulation, fp32 master '
                         f'weights) and has no fp32/fp64-storage mode; set trainer.precision=bf16 to run this recipe on it')
    raise ValueError(f'trainer.precision={p!r} is not a Lightning precision')


def resolve_strategy(trainer: Dict[str, Any]):
    """(distributed?, devices) from trainer.strategy / trainer.devices / the torchrun environment."""
    strat = str(trainer.get('strategy', 'auto'))
    if strat not in ('auto', 'ddp', 'ddp_find_unused_parameters_false', 'ddp_find_unused_parameters_true', 'ddp_spawn'):
        raise ValueError(f'trainer.strategy={strat!r}: the hot path is data-parallel only (ddp)')
    dev = trainer.get('devices', 'auto')
    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if isinstance(dev, (list, tuple)):
        n = len(dev)
    elif dev in ('auto', -1, '-1'):
        n = world_env if world_env > 1 else 1
    else:
        n = int(dev)
    for key, bad in (('accumulate_grad_batches', lambda v: v not in (None, 1)), ('gradient_clip_val', lambda v: v),
                     ('sync_batchnorm', lambda v: bool(v))):
        if bad(trainer.get(key)):
            raise NotImplementedError(f'trainer.{key}={trainer.get(key)!r} is not built on the hot path')
    return (n > 1 or world_env > 1), n, strat == 'ddp_find_unused_parameters_true'


def synthetic_batches(cfg: ConfigDict, batch_size: int, device, seed: int = 0):
    """Endless synthetic batches with the batch-dict contract of the reference datasets (keys image / target / index,
    examples/cifar.py:131-150; anchor / positive / negative for the triplet task), shaped by task.params.inputs."""
    tp = cfg.task.params
    shape = list(tp['inputs'][0]['shape'])
    g = torch.Generator(device='cpu').manual_seed(seed)
    name = cfg.task.name
    classes = int((tp.get('head_params') or {}).get('num_classes', 0) or 10)
    while True:
        def img():
            return torch.randn(batch_size, *shape, generator=g).to(device=device, dtype=torch.bfloat16)
        if name == 'TripletLearnTask':
            yield {'anchor': img(), 'positive': img(), 'negative': img()}
        elif name == 'SegmentationTask':
            yield {'image': img(), 'target': torch.randint(0, classes, (batch_size, *shape[1:]), generator=g).to(device)}
        else:
            yield {'image': img(), 'target': torch.randint(0, max(classes, 2), (batch_size,), generator=g).to(device),
                   'index': torch.arange(batch_size, device=device)}


def fit(cfg: ConfigDict, batches: Optional[Iterable[Dict[str, torch.Tensor]]] = None, max_steps: Optional[int] = None,
        batch_size: int = 8, device: Optional[str] = None, on_step: Optional[Callable] = None) -> Dict[str, Any]:
    """YAML config -> Task -> optimizer / scheduler -> (reducer) -> steps.  Returns the task, the last step's outputs,
    the step count and, on a multi-rank job, whether the replicas ended bit-identical."""
    import torch.distributed as dist
    trainer = cfg.get('trainer') or {}
    resolve_precision(trainer)
    distributed, devices, find_unused = resolve_strategy(trainer)
    if cfg.get('seed_params'):
        torch.manual_seed(int(cfg.seed_params.get('seed', 0)))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if device is None:
        device = f'cuda:{local_rank}'
    if str(device).startswith('cuda'):
        torch.cuda.set_device(device)
    if distributed and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl' if str(device).startswith('cuda') else 'gloo', rank=rank,
                                world_size=int(os.environ.get('WORLD_SIZE', devices)))
    task = TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(device).train()
    confs = task.configure_optimizers()
    if len(confs) != 1:
        raise NotImplementedError('one optimizer per task on the hot path (manual multi-optimizer loops are not built)')
    opt, sched = confs[0]['optimizer'], confs[0].get('lr_scheduler')
    reducer = None
    if distributed:
        from .dist import GradientAllReducer
        reducer = GradientAllReducer(opt, module=task, find_unused_parameters=find_unused)
    steps = max_steps if max_steps is not None else int(trainer.get('max_steps', -1) or -1)
    limit = trainer.get('limit_train_batches')
    epochs = int(trainer.get('max_epochs') or 1)
    per_epoch = int(limit) if isinstance(limit, int) and limit > 0 else None
    if steps is None or steps < 0:
        if per_epoch is None:
            raise ValueError('fit(): give max_steps (or trainer.max_steps, or an integer trainer.limit_train_batches with '
                             'trainer.max_epochs) — synthetic batches never run out')
        steps = per_epoch * epochs
    it = iter(batches) if batches is not None else synthetic_batches(cfg, batch_size, device, seed=1234 + rank)
    out, done = None, 0
    for i in range(steps):
        try:
            batch = next(it)
        except StopIteration:
            break
        out = train_step(task, opt, batch, i, reducer)
        done += 1
        if sched is not None and sched['interval'] == 'step' and (i + 1) % sched['frequency'] == 0 \
                and type(sched['scheduler']).__name__ != 'ReduceLROnPlateau':
            sched['scheduler'].step()
        epoch_end = per_epoch is not None and (i + 1) % per_epoch == 0
        if epoch_end or i == steps - 1:
            task.on_train_epoch_end()
            if sched is not None and sched['interval'] == 'epoch' and type(sched['scheduler']).__name__ != 'ReduceLROnPlateau':
                sched['scheduler'].step()
            task.current_epoch += 1
        if on_step is not None:
            on_step(i, out)
    result = {'task': task, 'optimizer': opt, 'steps': done, 'outputs': out, 'logged': dict(task.logged),
              'ranks_in_sync': replicas_in_sync(reducer), 'world': dist.get_world_size() if distributed else 1}
    if reducer is not None:
        reducer.close()
    return result


def _parse_override(s: str):
    import yaml
    k, _, v = s.partition('=')
    return k, yaml.safe_load(v)


def _spawned(rank, world, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    main(argv, _child=True)


def main(argv=None, _child=False):
    ap = argparse.ArgumentParser(prog='python -m torchok_amd.run', description=__doc__.split('\n\n')[0])
    ap.add_argument('-cp', '--config-path', required=True)
    ap.add_argument('-cn', '--config-name', required=True)
    ap.add_argument('--batch-size', type=int, default=8, help='synthetic batch size per GPU')
    ap.add_argument('overrides', nargs='*', help='dotted.key=value (the launcher-style overrides of the reference)')
    a = ap.parse_args(argv)
    name = a.config_name if a.config_name.endswith(('.yaml', '.yml')) else a.config_name + '.yaml'
    ov = dict(_parse_override(o) for o in a.overrides)
    mode = ov.pop('mode', 'train')
    if mode != 'train':
        raise ValueError(f'Entrypoint <{mode}>: only the training hot path is built (reference modes: train, test, predict, find_lr)')
    cfg = load_config(os.path.join(a.config_path, name), overrides=ov)
    distributed, devices, _ = resolve_strategy(cfg.get('trainer') or {})
    if distributed and 'RANK' not in os.environ and not _child:
        import torch.multiprocessing as mp
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        mp.spawn(_spawned, args=(devices, list(argv if argv is not None else sys.argv[1:])), nprocs=devices, join=True)
        return
    res = fit(cfg, batch_size=a.batch_size)
    if int(os.environ.get('RANK', '0')) == 0:
        loss = res['outputs']['loss'] if res['outputs'] else None
        print(f"fit: {res['steps']} steps on {res['world']} rank(s), last loss {float(loss) if loss is not None else None}, "
              f"ranks_in_sync {res['ranks_in_sync']}")


try:   # Lightning adapter: defined only where pytorch_lightning exists (absent on both boxes)
    import pytorch_lightning as _pl

    class LitTask(_pl.LightningModule):
        """A torchok_amd Task behind Lightning's module API: Trainer.fit drives the same hooks as `fit` above.  The
        gradient exchange stays with Lightning's DDP strategy in that case (parameters are ordinary nn.Parameters)."""

        def __init__(self, task):
            super().__init__()
            self.task = task
            task.log, task.log_dict = self.log, self.log_dict

        def forward(self, *args, **kwargs):
            return self.task(*args, **kwargs)

        def training_step(self, batch, batch_idx):
            return self.task.training_step(batch, batch_idx)

        def validation_step(self, batch, batch_idx, dataloader_idx=0):
            return self.task.validation_step(batch, batch_idx, dataloader_idx)

        def test_step(self, batch, batch_idx, dataloader_idx=0):
            return self.task.test_step(batch, batch_idx, dataloader_idx)

        def predict_step(self, batch, batch_idx, dataloader_idx=0):
            return self.task.predict_step(batch, batch_idx, dataloader_idx)

        def configure_optimizers(self):
            return self.task.configure_optimizers()

        def on_train_batch_end(self, outputs, batch, batch_idx):
            return self.task.on_train_batch_end(outputs, batch, batch_idx)

        def on_train_epoch_end(self):
            return self.task.on_train_epoch_end()

        def on_validation_epoch_end(self):
            return self.task.on_validation_epoch_end()
except ImportError:   # pragma: no cover - the only branch reachable on these boxes
    LitTask = None


if __name__ == '__main__':
    main()
