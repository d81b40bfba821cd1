"""The step function bench.py / run.py / GraphedTrainingStep share (torchok_amd/engine/step.py), on two gloo ranks with the
host stand-in for the kernels: training_step -> backward with the bucketed exchange -> optimizer step ->
on_train_batch_end, whose per-step loss mean (reference tasks/base.py:163-173) is issued asynchronously through the
reducer and consumed one step later.  After three steps the replicas must be bit-identical (`replicas_in_sync`)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torchok_amd as T
    from torchok_amd.dist import GradientAllReducer
    from torchok_amd.engine.step import replicas_in_sync, train_step
    from helpers import cls_config, deterministic_state
    import fake_backend as fb
    fb.install()
    cfg = cls_config('resnet18', 10)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    task.load_state_dict(deterministic_state(task.state_dict(), 3 + rank))      # ranks start apart: rank 0 is broadcast
    opt = task.configure_optimizers()[0]['optimizer']
    red = GradientAllReducer(opt, bucket_bytes=4 << 20, module=task)
    assert task._grad_reducer is red and not red.find_unused
    g = torch.Generator().manual_seed(300 + rank)
    losses = []
    for i in range(3):
        x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
        out = train_step(task, opt, {'image': x, 'target': y}, i, red)
        mine = out['loss'].detach().float().clone()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        losses.append(float(sum(both) / world))
        if i > 0:
            # the value logged after step i is the rank-mean of step i-1 (issued then, consumed now)
            assert abs(float(task.logged['train/loss']) - losses[i - 1]) < 1e-6
    flushed = task.flush_step_logs()
    assert abs(float(flushed['loss']) - losses[-1]) < 1e-6 and abs(float(task.logged['train/loss']) - losses[-1]) < 1e-6
    assert task.flush_step_logs() == {}
    assert replicas_in_sync(red) is True
    # a replica that drifts is detected
    with torch.no_grad():
        if rank == 1:
            next(iter(task.parameters())).add_(1e-3)
    assert replicas_in_sync(red) is False
    red.close()
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


@pytest.mark.timeout(300)
def test_shared_train_step_on_two_gloo_ranks(tmp_path):
    port = 33500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')


def test_single_rank_step_logs_immediately(fake_backend):
    """Without a process group the mean over ranks is the value itself and is logged in the same call."""
    import torchok_amd as T
    from torchok_amd.engine.step import replicas_in_sync, train_step
    from helpers import cls_config
    cfg = cls_config('resnet18', 10)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    opt = task.configure_optimizers()[0]['optimizer']
    out = train_step(task, opt, {'image': torch.randn(4, 3, 32, 32), 'target': torch.randint(0, 10, (4,))}, 0)
    assert abs(float(task.logged['train/loss']) - float(out['loss'])) < 1e-6
    assert replicas_in_sync(None) is None
