"""BCEWithLogitsLoss with an ignore value, on the streaming BCE kernels (the multi-label loss the reference registers in
``torchok/losses/classification/binary_cross_entropy.py:13-59``).  Elements whose target equals ``ignore_index`` do not
count (reference forward, :50-59); fp32 math on bf16 logits; 'mean' over the selected elements or 'sum'; a batch with
nothing selected yields 0.  The selection never leaves the device (the reference builds a boolean-masked copy)."""
import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from .cross_entropy import materialized
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr


class _BCELogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, ignore_value: float, mean: bool):
        require_device(logits)
        classes = logits.shape[-1] if logits.dim() > 1 else 1
        z = logits.detach().reshape(-1, classes)
        if z.dtype != BF16 or z.stride(1) != 1 or z.stride(0) < classes:
            z = z.to(BF16).contiguous()
        rows = z.shape[0]
        tgt = target.detach().reshape(rows, classes).to(torch.float32).contiguous()
        loss = torch.empty(_C.TOK_CE_LOSS_FLOATS, dtype=torch.float32, device=z.device)
        _C.check(_C.lib().tok_bce_logits_fwd(ptr(z), ptr(tgt), rows, classes, z.stride(0), float(ignore_value), int(mean),
                                             ptr(loss), stream_ptr()), 'tok_bce_logits_fwd')
        ctx.z, ctx.tgt, ctx.loss, ctx.ignore_value, ctx.mean = z, tgt, loss, float(ignore_value), int(mean)
        ctx.in_shape, ctx.in_dtype = tuple(logits.shape), logits.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        z = ctx.z
        rows, classes = z.shape
        ld = z.stride(0)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty((rows, ld), dtype=BF16, device=z.device)
        _C.check(_C.lib().tok_bce_logits_bwd(ptr(z), ptr(ctx.tgt), ptr(ctx.loss), ptr(gs), rows, classes, ld,
                                             ctx.ignore_value, ctx.mean, ptr(d), stream_ptr()), 'tok_bce_logits_bwd')
        ctx.z = ctx.tgt = ctx.loss = None
        if ld != classes:
            if ld == pad8(classes):
                mark_padded(d)
            d = d[:, :classes]
        d = d.reshape(ctx.in_shape)
        if ctx.in_dtype != BF16:
            d = d.to(ctx.in_dtype)
        return d, None, None, None


@LOSSES.register_class
class BCEWithLogitsLoss(nn.Module):
    def __init__(self, weight=None, reduction: str = 'mean', pos_weight=None, ignore_index: int = -1):
        super().__init__()
        if weight is not None or pos_weight is not None:
            # the reference flattens input and target through a boolean mask before the functional call (:52-53), so a
            # per-class `pos_weight` / per-sample `weight` only broadcasts there by accident of the selected count
            raise NotImplementedError('torchok_amd BCEWithLogitsLoss: weight / pos_weight are not supported')
        if reduction not in ('mean', 'sum'):
            raise NotImplementedError("torchok_amd BCEWithLogitsLoss: reduction 'mean' or 'sum' (\"none\" has a "
                                      "data-dependent shape in the reference)")
        self.reduction = reduction
        self.ignore_index = ignore_index

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        input = materialized(input)   # an UpsampledLogits would hide the autograd edge from Function.apply
        if tuple(input.shape) != tuple(target.shape):
            raise ValueError(f'BCEWithLogitsLoss: input {tuple(input.shape)} and target {tuple(target.shape)} differ')
        return _BCELogits.apply(input, target, float(self.ignore_index), self.reduction == 'mean')
