"""Two ranks of the REAL HIP path on ONE GPU (reference: Lightning `strategy: ddp`,
constructor/runner.py:18, and BaseTask.on_train_batch_end, tasks/base.py:163-173).

RCCL needs two devices, but torch's gloo backend all-reduces device tensors (staged through the host), so two processes
that share `cuda:0` exercise everything of the N > 1 path except the wire: the HIP kernels, gradients arriving from the
engine's side stream, the bucket events, the comm stream, the flat buffer broadcast, the asynchronous loss mean and — for
the transformer backbone — `find_unused_parameters` with its cached used-map (`static_unused_pattern=True`).  Cases: ResNet-18
(BatchNorm, strict mode, fp32 payload), a small SwinV2, ResNet-18 with the bf16 gradient payload (`grad_dtype='bf16'`), and
the HRNet segmentation task, whose backward writes into the buckets from the branch streams (the case the reducer's
per-bucket stream list exists for).  Checked per model:

 * the exchanged gradient equals the mean of the two ranks' local gradients (local BatchNorm statistics),
 * after three shared `train_step`s parameters AND buffers are bit-identical on both ranks,
 * `finish_step` issues no host synchronisation in the steady state (no `.cpu()` / `.item()` / `synchronize()` from the
   reducer's Python; gloo's own host staging is outside that claim, RCCL has none)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class _SyncCounter:
    """Counts the torch-level calls that block the host on the device while active."""
    def __init__(self):
        self.calls = []
        self._saved = []

    def _wrap(self, owner, name):
        real = getattr(owner, name)
        counter = self

        def spy(*a, **k):
            counter.calls.append(f'{getattr(owner, "__name__", owner)}.{name}')
            return real(*a, **k)
        self._saved.append((owner, name, real))
        setattr(owner, name, spy)

    def __enter__(self):
        for owner, name in ((torch.Tensor, 'cpu'), (torch.Tensor, 'item'), (torch.Tensor, 'tolist'), (torch.Tensor, 'numpy'),
                            (torch.cuda, 'synchronize'), (torch.cuda.Stream, 'synchronize'),
                            (torch.cuda.Event, 'synchronize')):
            self._wrap(owner, name)
        return self

    def __exit__(self, *exc):
        for owner, name, real in self._saved:
            setattr(owner, name, real)
        self._saved = []
        return False


def _worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torchok_amd as T
    from torchok_amd.dist import GradientAllReducer
    from torchok_amd.engine.step import replicas_in_sync, train_step
    from helpers import cls_config, deterministic_state
    dev = 'cuda:0'

    def build(backbone, seed, optimizer, opt_params, **bk):
        if backbone.startswith('hrnet'):
            from test_hrnet import seg_config
            cfg = seg_config(backbone, classes=5, size=64)
        else:
            cfg = cls_config(backbone, 10, optimizer=optimizer, opt_params=opt_params, backbone_params=bk or None,
                             inputs_shape=(3, 64, 64))
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')},
                                 seed + 100 * rank)
        task.load_state_dict(sd, strict=False)       # ranks start apart: the reducer must broadcast rank 0's weights
        return task.to(dev).train()

    cases = [
        ('resnet18', dict(), 'SGD', {'lr': 0.05, 'momentum': 0.9, 'weight_decay': 1e-4}, False, 'fp32'),
        ('swinv2_custom', dict(img_size=64, window_size=4, depths=[2, 2, 2, 2], drop_path_rate=0.0), 'AdamW',
         {'lr': 1e-3, 'weight_decay': 0.05}, True, 'fp32'),
        ('resnet18', dict(), 'SGD', {'lr': 0.05, 'momentum': 0.9, 'weight_decay': 1e-4}, False, 'bf16'),
        ('hrnet_w18_small', dict(), 'SGD', {}, False, 'fp32'),
    ]
    for backbone, bk, oname, oparams, find_unused, payload in cases:
        task = build(backbone, 3, oname, oparams, **bk)
        opt = task.configure_optimizers()[0]['optimizer']
        red = GradientAllReducer(opt, bucket_bytes=1 << 20, module=task, find_unused_parameters=find_unused,
                                 static_unused_pattern=True if find_unused else None, grad_dtype=payload)
        assert task._grad_reducer is red and red.bf16 == (payload == 'bf16')
        g = torch.Generator().manual_seed(50 + rank)
        x = torch.randn(16, 3, 64, 64, generator=g).to(dev)
        if backbone.startswith('hrnet'):
            y = torch.randint(0, 5, (16, 64, 64), generator=g).to(dev)
        else:
            y = torch.randint(0, 10, (16,), generator=g).to(dev)
        batch = {'image': x, 'target': y}

        # (1) mean gradient == mean of the two local gradients
        out = task.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        out['loss'].backward()                         # local gradients, no exchange
        with_grad = [p.grad is not None for p in task.parameters()]
        local = torch.cat([p.grad.flatten() for p in task.parameters() if p.grad is not None]).clone()
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        want = sum(both) / world
        out = task.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        red.begin_step()
        out['loss'].backward()
        red.finish_step()
        assert [p.grad is not None for p in task.parameters()] == with_grad      # unused everywhere stays None
        if find_unused:
            assert not all(with_grad)                  # the per-stage feature norms of a transformer backbone
        got = torch.cat([p.grad.flatten() for p in task.parameters() if p.grad is not None])
        # fp32 payload: the mean itself; bf16 payload: each bucket is narrowed before the exchange (torch's bf16_compress_hook)
        assert float((got - want).norm() / want.norm()) < (1e-5 if payload == 'fp32' else 2 ** -7), (backbone, payload)
        assert len(red.buckets[0]) > 1                 # several buckets: launched from the gradient hooks during backward
        # gradients reach a bucket from more than one stream (weight gradients on the side stream, BatchNorm / bias gradients
        # on the main one, HRNet's branches on theirs): the exchange waited for each of them
        assert max(len(b.streams) for b in red.buckets[0]) >= 2, (backbone, [len(b.streams) for b in red.buckets[0]])
        opt.zero_grad(set_to_none=True)

        # (2) three steps of the shared train_step; no host synchronisation inside finish_step once the pattern is known
        real_finish = red.finish_step
        sync_calls = []

        def spied_finish():
            with _SyncCounter() as c:
                real_finish()
            sync_calls.append(list(c.calls))
        red.finish_step = spied_finish
        red.enable_timing()          # bench.py's N > 1 diagnosis: two timing events per finish_step, no host synchronisation
        for i in range(3):
            train_step(task, opt, batch, i, red)
        red.finish_step = real_finish
        assert sync_calls[1] == [] and sync_calls[2] == [], (backbone, sync_calls)
        if not find_unused:
            assert sync_calls[0] == [], (backbone, sync_calls)
        torch.cuda.synchronize()
        exposed = red.exposed_comm_ms()      # step-stream time between "backward done" and "exchange joined", per step
        assert len(exposed) == 3 and all(0.0 <= v < 5e3 for v in exposed), (backbone, exposed)
        red.enable_timing(False)
        assert red.exposed_comm_ms() == []
        flat = torch.cat([p.detach().flatten() for p in task.parameters()] +
                         [b.detach().float().flatten() for b in task.buffers()])
        every = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(every, flat)
        assert torch.equal(every[0], every[1]), backbone     # parameters AND buffers identical on both ranks
        assert replicas_in_sync(red) is True
        logged = task.flush_step_logs()
        assert 'loss' in logged and bool(torch.isfinite(logged['loss']))
        red.close()
        assert not hasattr(task, '_grad_reducer')
        dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


@pytest.mark.timeout(600)
def test_two_processes_on_one_gpu_over_gloo(tmp_path):
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
