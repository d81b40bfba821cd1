"""GradientAllReducer on the GPU with a ONE-rank RCCL group (the only multi-process-free way to run the real
stream / event / RCCL plumbing on a 1-GPU box): bucketed all-reduce launched from the gradient hooks — weight
gradients arrive from the engine's side stream, BatchNorm gradients from the main one — must leave exactly the
gradients and parameters of a run without the reducer.  (The 2-rank arithmetic is covered on CPU/gloo.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import torchok_amd as T
from helpers import cls_config, deterministic_state

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_single_rank_rccl_reducer_is_transparent():
    assert torch.cuda.is_available()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        from torchok_amd.dist import GradientAllReducer
        results = []
        for use_reducer in (False, True):
            cfg = cls_config('resnet18', 10)
            task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
            sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 3)
            task.load_state_dict(sd, strict=False)
            task.cuda().train()
            opt = task.configure_optimizers()[0]['optimizer']
            red = GradientAllReducer(opt, bucket_bytes=4 << 20, module=task) if use_reducer else None
            g = torch.Generator().manual_seed(0)
            x = torch.randn(16, 3, 64, 64, generator=g).cuda()
            y = torch.randint(0, 10, (16,), generator=g).cuda()
            for it in range(3):
                out = task.training_step({'image': x, 'target': y}, it)
                opt.zero_grad(set_to_none=True)
                if red is not None:
                    red.begin_step()
                out['loss'].backward()
                if red is not None:
                    red.finish_step()
                opt.step()
            torch.cuda.synchronize()
            results.append({n: p.detach().clone() for n, p in task.named_parameters()})
            if red is not None:
                assert len(red.buckets[0]) > 1          # the arena really was cut into several buckets
                red.close()
        for n in results[0]:
            assert torch.equal(results[0][n], results[1][n]), n     # wgrad / reduce kernels are deterministic
    finally:
        dist.destroy_process_group()


# ---- two ranks over RCCL / xGMI (self-skipping on a 1-GPU box) ----------------------------------------------------

def _rccl_worker(rank, world, port, tmp, grad_dtype):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    import torchok_amd as T2
    from torchok_amd.dist import GradientAllReducer
    from helpers import cls_config as cc, deterministic_state as ds
    dev = f'cuda:{rank}'

    def build(backbone, seed, **bk):
        cfg = cc(backbone, 10, backbone_params=bk or None, inputs_shape=(3, 64, 64))
        task = T2.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = ds({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, seed + 100 * rank)
        task.load_state_dict(sd, strict=False)       # ranks start apart: the reducer must broadcast rank 0's weights
        return task.to(dev).train()

    # (1) BatchNorm model: per-rank (local) batch statistics, mean gradient == mean of the two local gradients
    task = build('resnet18', 3)
    opt = task.configure_optimizers()[0]['optimizer']
    red = GradientAllReducer(opt, bucket_bytes=4 << 20, module=task, grad_dtype=grad_dtype)
    g = torch.Generator().manual_seed(50 + rank)
    x = torch.randn(16, 3, 64, 64, generator=g).to(dev)
    y = torch.randint(0, 10, (16,), generator=g).to(dev)
    out = task.training_step({'image': x, 'target': y}, 0)
    opt.zero_grad(set_to_none=True)
    out['loss'].backward()                         # local gradients, no exchange
    local = torch.cat([p.grad.flatten() for p in task.parameters()]).clone()
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    want = sum(both) / world
    bn_before = task.backbone.bn1.running_mean.clone()
    out = task.training_step({'image': x, 'target': y}, 0)
    opt.zero_grad(set_to_none=True)
    red.begin_step()
    out['loss'].backward()
    red.finish_step()
    got = torch.cat([p.grad.flatten() for p in task.parameters()])
    tol = 2 ** -7 if grad_dtype == 'bf16' else 1e-5
    assert float((got - want).norm() / want.norm()) < tol
    opt.step()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for p in task.parameters()] +
                     [b.detach().float().flatten() for b in task.buffers()])
    every = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(every, flat)
    assert torch.equal(every[0], every[1])         # parameters AND buffers (BN running stats) identical on both ranks
    assert not torch.equal(task.backbone.bn1.running_mean, bn_before)
    red.close()
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


@pytest.mark.parametrize('grad_dtype', ['fp32', 'bf16'])
def test_two_rank_rccl_mean_gradient(tmp_path, grad_dtype):
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (a multi-GPU node); the 2-rank arithmetic runs on gloo in test_ddp_gloo*.py')
    import torch.multiprocessing as mp
    mp.spawn(_rccl_worker, args=(2, _free_port(), str(tmp_path), grad_dtype), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
