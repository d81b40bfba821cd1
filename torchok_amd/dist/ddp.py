"""Data-parallel gradient exchange: bucketed mean all-reduce of the flat gradient arena over
RCCL (xGMI) on a side HIP stream, overlapped with backward.

What it replaces: Lightning's ``strategy: ddp`` = torch DistributedDataParallel's C++ reducer
(25 MiB buckets, NCCL all-reduce overlapped with backward) that the reference
enables through ``trainer.strategy`` (``torchok/constructor/runner.py:18``).

MI355X design: one process per GPU; gradients already live in ONE contiguous fp32 arena in
registration order (engine/arena.py), so a bucket is simply an arena range — no copy-in /
copy-out.  Gradients become ready from the END of the arena during backward; buckets are cut
from the end, each is launched on the comm stream as soon as its last parameter gradient has been
written (hipEvent dependency from the compute stream), and the optimizer waits on the comm
stream.  xGMI is point-to-point (7 links x ~153 GB/s): few LARGE messages amortise the per-link
ring latency, hence arena-range buckets of >= 32 MiB instead of per-parameter messages.

The path shards by images only (pure data parallelism): no activation exchange, BatchNorm
statistics stay per-GPU (the reference default, config_structure.py:170 sync_batchnorm=False).
Deviation: BN running-statistic buffers are NOT re-broadcast from rank 0 before every forward
(DDP broadcast_buffers=True); each rank keeps the running stats of its own shard.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _C
from ..engine import core
from ..engine.arena import ParamArena
from ..engine.core import ptr, stream_ptr


class _Bucket:
    __slots__ = ('lo', 'hi', 'first', 'last', 'pending', 'work', 'streams')

    def __init__(self, lo, hi, first, last):
        self.lo, self.hi, self.first, self.last = lo, hi, first, last
        self.pending = 0
        self.work = None
        self.streams = {}     # streams that produced gradients of this bucket in the current step


class GradientAllReducer:
    def __init__(self, optimizer, bucket_bytes: int = 32 << 20, process_group=None, broadcast_params: bool = True):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        optimizer._ensure_built()
        self.arenas: List[ParamArena] = [a for a in optimizer._arenas if a is not None]
        self.cuda = self.arenas[0].master.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.cuda else None
        self.buckets: List[List[_Bucket]] = []
        self._owner = {}
        for ai, arena in enumerate(self.arenas):
            blist = []
            hi_i = len(arena.params) - 1
            hi = arena.total
            i = hi_i
            while i >= 0:
                lo = arena.offsets[i]
                if (hi - lo) * 4 >= bucket_bytes or i == 0:
                    blist.append(_Bucket(lo, hi, i, hi_i))
                    hi, hi_i = lo, i - 1
                i -= 1
            self.buckets.append(blist)
            for b in blist:
                for pi in range(b.first, b.last + 1):
                    self._owner[id(arena.params[pi])] = (ai, b)
            if broadcast_params:
                dist.broadcast(arena.master, src=0, group=process_group)
        self._avg = dist.ReduceOp.AVG if (self.cuda and dist.get_backend(process_group) == 'nccl') else None
        self._active = False
        core.param_grad_hooks.append(self._on_grad)

    # ---- per step ---------------------------------------------------------------------------
    def begin_step(self):
        """Call before backward: arms the buckets."""
        for ai, blist in enumerate(self.buckets):
            arena = self.arenas[ai]
            for b in blist:
                b.pending = sum(1 for pi in range(b.first, b.last + 1) if arena.params[pi].requires_grad)
                b.work = None
                b.streams = {}
        self._active = True

    def _on_grad(self, p):
        if not self._active:
            return
        ent = self._owner.get(id(p))
        if ent is None:
            return
        ai, b = ent
        if self.cuda:
            # weight gradients are produced on the engine's side stream, BatchNorm / bias gradients on the main one:
            # the exchange has to wait for every stream that wrote into the bucket
            cur = torch.cuda.current_stream()
            b.streams[cur.cuda_stream] = cur
        b.pending -= 1
        if b.pending == 0:
            self._launch(ai, b)

    def _launch(self, ai: int, b: _Bucket):
        view = self.arenas[ai].grad[b.lo:b.hi]
        if self.cuda:
            cur = torch.cuda.current_stream()
            b.streams[cur.cuda_stream] = cur
            for s_ in b.streams.values():
                ev = torch.cuda.Event()
                ev.record(s_)
                self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self._avg is not None:
                    b.work = dist.all_reduce(view, op=self._avg, group=self.group, async_op=True)
                else:
                    b.work = dist.all_reduce(view, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(view, group=self.group, async_op=True)

    def finish_step(self):
        """Call after backward, before optimizer.step(): flush stragglers, join the comm stream."""
        for ai, blist in enumerate(self.buckets):
            for b in blist:
                if b.work is None:
                    self._launch(ai, b)
        for ai, blist in enumerate(self.buckets):
            for b in blist:
                b.work.wait()   # on CUDA: makes the CURRENT stream wait for the collective (no host sync)
                if self._avg is None:
                    view = self.arenas[ai].grad[b.lo:b.hi]
                    if self.cuda or _C.is_fake():
                        _C.check(_C.lib().tok_scale_f32(ptr(view), 1.0 / self.world, view.numel(), stream_ptr()),
                                 'tok_scale_f32')
                    else:
                        view.mul_(1.0 / self.world)
        self._active = False

    def close(self):
        if self._on_grad in core.param_grad_hooks:
            core.param_grad_hooks.remove(self._on_grad)
