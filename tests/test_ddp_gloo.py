"""N > 1 path on CPU: two processes over gloo, the native library replaced by the host stand-in.
Checks that the bucketed arena all-reduce gives every rank the MEAN gradient (what Lightning-DDP
does for the reference) and that both ranks stay bit-identical after the step."""
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
    from torchok_amd import _C
    from torchok_amd.dist import GradientAllReducer
    from helpers import cls_config, deterministic_state
    import fake_backend as fb
    fb.install()
    cfg = cls_config('resnet18', 10)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    # rank 1 starts from different weights: the reducer must broadcast rank 0's
    task.load_state_dict(deterministic_state(task.state_dict(), 7 + rank))
    opt = task.configure_optimizers()[0]['optimizer']
    red = GradientAllReducer(opt, bucket_bytes=4 << 20)
    assert len(red.buckets[0]) > 3
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    out = task.training_step({'image': x, 'target': y}, 0)
    opt.zero_grad()
    # local gradients first (no exchange) for the reference mean
    out['loss'].backward()
    local = torch.cat([p.grad.flatten() for p in task.parameters()]).clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    # now the same step with the reducer armed
    out = task.training_step({'image': x, 'target': y}, 0)
    opt.zero_grad()
    red.begin_step()
    out['loss'].backward()
    red.finish_step()
    got = torch.cat([p.grad.flatten() for p in task.parameters()])
    # second forward saw updated BN running stats only; batch-stat gradients are identical
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), float((got - want).abs().max())
    opt.step()
    flat = torch.cat([p.detach().flatten() for p in task.parameters()])
    others = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(others, flat)
    assert torch.equal(others[0], others[1])
    # logged losses: mean over ranks from one collective (reference: all_gather + mean, tasks/base.py:163-173)
    mine = out['loss'].detach().float()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    logged = task.on_train_batch_end({'loss': out['loss']}, None, 0)
    assert set(logged) == {'loss'} and abs(float(logged['loss']) - float(sum(both) / world)) < 1e-6
    assert 'train/loss' in task.logged
    red.close()
    # retrieval meter states are concatenated over ranks (torchmetrics dist_reduce_fx="cat",
    # metrics/index_base_metric.py:112-120; reference test: tests/base_tests/metrics/representation/
    # test_representation_ddp.py): uneven shards of the known-answer data give every rank the single-process value
    import numpy as np
    known = np.load(os.path.join(HERE, 'golden', 'retrieval_known_answers.npz'))
    sl = slice(0, 5) if rank == 0 else slice(5, 9)
    for k in (1, 3, 6):
        m = T.METRICS.get('RecallAtKMeter')(dataset_type='classification', normalize_vectors=True, k=k)
        m.update(vectors=torch.from_numpy(known['vectors'][sl]), group_labels=torch.from_numpy(known['targets'][sl]))
        assert abs(m.compute() - known['answer__classification__recall'][k - 1]) < 1e-6
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


@pytest.mark.timeout(300)
def test_two_rank_gloo_allreduce(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
