"""CrossEntropyLoss on the fused softmax-CE kernels (registered under the torch class name the
reference registers: ``torchok/losses/__init__.py:26``; called as ``loss(input=..., target=...)``
by JointLoss, ``losses/base.py:78-79``).  fp32 math on bf16 logits, mean reduction over the
non-ignored rows, like ``torch.nn.CrossEntropyLoss`` under bf16 autocast."""
import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, ignore_index: int):
        require_device(logits)
        rows, classes = logits.shape
        z = logits.detach()
        if z.dtype != BF16 or z.stride(1) != 1:
            z = z.to(BF16).contiguous()
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.to(torch.int64).contiguous()
        dev = z.device
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty(2, dtype=torch.float32, device=dev)
        _C.check(_C.lib().tok_softmax_ce_fwd(ptr(z), ptr(target), rows, classes, z.stride(0), ignore_index,
                                             ptr(lse), ptr(row_loss), ptr(loss), stream_ptr()),
                 'tok_softmax_ce_fwd')
        ctx.z, ctx.target, ctx.lse, ctx.loss, ctx.ignore_index = z, target, lse, loss, ignore_index
        ctx.in_dtype = logits.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        z, target = ctx.z, ctx.target
        rows, classes = z.shape
        ld = pad8(classes)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty((rows, ld), dtype=BF16, device=z.device)
        _C.check(_C.lib().tok_softmax_ce_bwd(ptr(z), ptr(target), ptr(ctx.lse), ptr(ctx.loss), ptr(gs), rows,
                                             classes, z.stride(0), ctx.ignore_index, ptr(d), stream_ptr()),
                 'tok_softmax_ce_bwd')
        ctx.z = ctx.target = ctx.lse = ctx.loss = None
        if ld != classes:
            mark_padded(d)
            d = d[:, :classes]
        if ctx.in_dtype != BF16:
            d = d.to(ctx.in_dtype)
        return d, None, None


@LOSSES.register_class
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, size_average=None, ignore_index: int = -100, reduce=None,
                 reduction: str = 'mean', label_smoothing: float = 0.0):
        super().__init__()
        if weight is not None or label_smoothing != 0.0 or reduction != 'mean' \
                or size_average is not None or reduce is not None:
            raise NotImplementedError('torchok_amd CrossEntropyLoss: mean reduction, no class weights, '
                                      'no label smoothing')
        self.ignore_index = ignore_index

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if input.dim() != 2:
            raise NotImplementedError('torchok_amd CrossEntropyLoss: (N, C) logits only (for now)')
        return _SoftmaxCE.apply(input, target, self.ignore_index)
