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
            red = GradientAllReducer(opt, bucket_bytes=4 << 20) if use_reducer else None
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
