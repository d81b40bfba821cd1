"""Which HIP stream / hardware queue does torch's ProcessGroupNCCL put a collective on?  One rank, RCCL backend; run under
rocprofv3 --kernel-trace and read the queue of oneRankReduce next to the marker kernels:
   phase A  all_reduce(async_op=True)  called with the picked comm stream current
   phase B  all_reduce(async_op=False) called with the picked comm stream current
Marker kernels: tok_cast_f32_bf16 on the comm stream right before each phase's collectives (so the comm stream's queue is known),
a fill on the default stream."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchok_amd import _C                      # noqa: E402
from torchok_amd.engine.core import pick_stream, ptr   # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
dist.init_process_group('nccl', rank=0, world_size=1)
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
comm = pick_stream(dev)
x = torch.ones(1 << 22, device=dev)
nb = torch.empty(1 << 22, dtype=torch.bfloat16, device=dev)
lib = _C.lib()
torch.cuda.synchronize()
for phase, async_op in (('A', True), ('B', False)):
    x.add_(1.0)                                   # default-stream marker
    ev = torch.cuda.Event()
    ev.record()
    comm.wait_event(ev)
    with torch.cuda.stream(comm):
        lib.tok_cast_f32_bf16(ptr(x), ptr(nb), x.numel(), comm.cuda_stream)      # comm-stream marker
        for _ in range(3 if async_op else 5):
            w = dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=async_op)      # (AVG: a one-rank group still launches a kernel)
            if w is not None:
                w.wait()
        lib.tok_cast_f32_bf16(ptr(x), ptr(nb), x.numel(), comm.cuda_stream)
    torch.cuda.current_stream().wait_stream(comm)
    torch.cuda.synchronize()
print('comm stream handle', hex(comm.cuda_stream), 'phase A: 3 async collectives, phase B: 5 sync collectives')
dist.destroy_process_group()
