"""Edges of the data-parallel exchange on two gloo ranks (host stand-in for the kernels):
 * a parameter whose gradient comes from plain torch autograd (outside the gradient arena) is still averaged,
 * a parameter used on ONE rank only gets the same (averaged) update on both — ranks never diverge,
 * a parameter unused everywhere keeps grad None (skipped by the optimizer, as under torch DDP),
 * module buffers follow rank 0 after every step (DDP broadcast_buffers=True, one flat broadcast per dtype),
 * bf16 gradient buckets (torch DDP's bf16_compress_hook) stay within bf16 rounding of the fp32 exchange,
 * classification metrics sum their counts over ranks at compute() (torchmetrics dist_reduce_fx='sum'),
 * a parameter group added after the first step joins the exchange,
 * without find_unused_parameters a missing gradient raises (torch DDP's behaviour)."""
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
    task.load_state_dict(deterministic_state(task.state_dict(), 11))
    q = torch.nn.Parameter(torch.tensor([1.0, -2.0]))     # torch-native gradient, used on both ranks
    r = torch.nn.Parameter(torch.tensor([0.5]))           # used on rank 1 only
    u = torch.nn.Parameter(torch.tensor([3.0]))           # used nowhere
    params = list(task.parameters()) + [q, r, u]
    opt = T.OPTIMIZERS.get('SGD')(params, lr=0.1, momentum=0.9)
    red = GradientAllReducer(opt, bucket_bytes=4 << 20, module=task, find_unused_parameters=True,
                             static_unused_pattern=True)      # the opt-in: host read only when the local pattern changes
    g = torch.Generator().manual_seed(200 + rank)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)

    def step(i, reducer, use_r=True):
        out = task.training_step({'image': x, 'target': y}, i)
        loss = out['loss'] + (q * torch.tensor([1.0 + rank, 2.0])).sum()
        if rank == 1 and use_r:
            loss = loss + 4.0 * r.sum()
        opt.zero_grad()
        reducer.begin_step()
        loss.backward()
        reducer.finish_step()
        return out

    step(0, red)
    assert torch.allclose(q.grad, torch.tensor([1.5, 2.0]))          # mean of (1, 2) and (2, 2)
    assert r.grad is not None and torch.allclose(r.grad, torch.tensor([2.0]))   # mean of (unused -> 0) and 4
    assert u.grad is None
    opt.step()
    flat = torch.cat([p.detach().flatten() for p in params])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    assert float(u) == 3.0
    # buffers: every rank holds rank 0's BatchNorm running statistics after the step
    bufs = torch.cat([b.detach().double().flatten() for b in task.buffers()])
    allb = [torch.zeros_like(bufs) for _ in range(world)]
    dist.all_gather(allb, bufs)
    assert torch.equal(allb[0], allb[1])
    nbt = [b for n, b in task.named_buffers() if n.endswith('num_batches_tracked')]
    assert all(int(b) == 1 for b in nbt)
    sd = task.state_dict()
    assert any(k.endswith('running_mean') for k in sd)
    red.enable_timing()                       # bench.py's N > 1 diagnosis needs HIP events: a no-op on the CPU backend
    assert red.exposed_comm_ms() == []

    # steady state of find_unused_parameters: this rank's pattern of missing gradients does not change, so the reduced used-map
    # is not read on the host again (one blocking copy per step under torch DDP) and the per-parameter scan does not run;
    # the averaged update still reaches `r` on rank 0 and the replicas stay identical
    reads, scans = [], []
    real_cpu, real_flags = torch.Tensor.cpu, GradientAllReducer._local_flags
    torch.Tensor.cpu = lambda t, *a, **k: (reads.append(1), real_cpu(t, *a, **k))[1]

    def counted_flags(self_):
        f, ch = real_flags(self_)
        scans.append(ch)
        return f, ch
    GradientAllReducer._local_flags = counted_flags
    for i in range(3):
        step(10 + i, red)
        assert r.grad is not None and torch.allclose(r.grad, torch.tensor([2.0])) and u.grad is None
        opt.step()
    torch.Tensor.cpu, GradientAllReducer._local_flags = real_cpu, real_flags
    assert reads == [] and scans == [False, False, False]
    flat = torch.cat([p.detach().flatten() for p in params])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    # a change on the OTHER rank that this rank cannot see (rank 1 stops using `r`; rank 0 never had a gradient for it) is
    # caught by the device-side comparison on rank 0; the verdict is MAX-reduced over the ranks, so BOTH ranks raise in the
    # next finish_step — together, before either launches a collective (with a per-rank verdict rank 0 raised
    # while rank 1 waited inside all_reduce)
    step(20, red, use_r=False)
    with pytest.raises(RuntimeError, match='static_unused_pattern'):
        red._poll_late_check()
    dist.barrier()
    red.close()

    # the DEFAULT (static_unused_pattern=False, torch DDP's per-step host read) follows a pattern that changes on the other
    # rank only: no error, `r` gets the averaged update while rank 1 uses it and grad None on both ranks once it does not
    snap_dyn = [p.detach().clone() for p in params]
    dyn = GradientAllReducer(opt, bucket_bytes=4 << 20, broadcast_params=False, find_unused_parameters=True)
    assert dyn.static_unused is False
    for i, use_r in enumerate((True, False, True, False, False)):
        step(30 + i, dyn, use_r=use_r)
        if use_r:
            assert r.grad is not None and torch.allclose(r.grad, torch.tensor([2.0])), (rank, i)
        else:
            assert r.grad is None, (rank, i)
        assert u.grad is None
        opt.step()
        flat = torch.cat([p.detach().flatten() for p in params])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), i
    dyn.close()
    with torch.no_grad():
        for p, sn in zip(params, snap_dyn):
            p.copy_(sn)
    red = GradientAllReducer(opt, bucket_bytes=4 << 20, module=task, find_unused_parameters=True)

    # bf16 buckets: same step from the same state, compared with the fp32 exchange
    red.close()
    assert not hasattr(task, '_grad_reducer')      # a closed reducer no longer receives the task's loss-mean collectives
    snap = [p.detach().clone() for p in params]
    step(1, GradientAllReducer(opt, bucket_bytes=4 << 20, broadcast_params=False, find_unused_parameters=True))
    g32 = torch.cat([p.grad.flatten() for p in params if p.grad is not None]).clone()
    red16 = GradientAllReducer(opt, bucket_bytes=4 << 20, broadcast_params=False, grad_dtype='bf16',
                               find_unused_parameters=True)
    step(1, red16)
    g16 = torch.cat([p.grad.flatten() for p in params if p.grad is not None])
    assert float((g16 - g32).abs().max()) <= float(g32.abs().max()) * 2 ** -7
    assert float((g16 - g32).norm() / g32.norm()) < 2 ** -8
    gg = [torch.zeros_like(g16) for _ in range(world)]
    dist.all_gather(gg, g16.clone())
    assert torch.equal(gg[0], gg[1])
    for p, s in zip(params, snap):
        assert torch.equal(p.detach(), s)

    # a parameter group added after the first step is exchanged too
    late = torch.nn.Parameter(torch.tensor([1.0, 1.0, 1.0]))
    opt.add_param_group({'params': [late]})
    params.append(late)
    out = task.training_step({'image': x, 'target': y}, 2)
    loss = out['loss'] + (late * float(rank + 1)).sum() + q.sum()
    opt.zero_grad()
    red16.begin_step()
    loss.backward()
    red16.finish_step()
    assert torch.allclose(late.grad, torch.full((3,), 1.5))
    opt.step()
    assert torch.allclose(late.detach(), torch.full((3,), 1.0 - 0.1 * 1.5))
    red16.close()

    # the default is torch DDP's find_unused_parameters=False: a parameter without a gradient is an error, not a silent
    # divergence (`u` is unused on both ranks, so both raise before any straggler bucket is exchanged)
    strict = GradientAllReducer(opt, bucket_bytes=4 << 20, broadcast_params=False)
    with pytest.raises(RuntimeError, match='find_unused_parameters'):
        step(3, strict)
    strict.close()

    # metrics: per-rank shards, one value
    acc = T.METRICS.get('Accuracy')(task='multiclass', num_classes=3)
    preds = torch.tensor([[9., 0, 0], [0, 9., 0], [0, 0, 9.], [9., 0, 0]])
    target = torch.tensor([0, 1, 2, 0]) if rank == 0 else torch.tensor([1, 2, 0, 1])     # rank 0: 4/4, rank 1: 0/4
    acc.update(preds, target)
    assert abs(float(acc.compute()) - 0.5) < 1e-6
    mae = T.METRICS.get('MeanAbsoluteError')()
    mae.update(torch.full((4,), float(rank)), torch.zeros(4))
    assert abs(float(mae.compute()) - 0.5) < 1e-6
    dist.destroy_process_group()
    open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')


@pytest.mark.timeout(300)
def test_two_rank_gloo_edges(tmp_path):
    port = 31500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / 'ok0') and os.path.exists(tmp_path / 'ok1')
