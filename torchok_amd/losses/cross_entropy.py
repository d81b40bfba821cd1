"""CrossEntropyLoss on the fused softmax-CE kernels (registered under the torch class name the
reference registers: ``torchok/losses/__init__.py:26``; called as ``loss(input=..., target=...)``
by JointLoss, ``losses/base.py:78-79``).  fp32 math on bf16 logits, mean (or sum) over the
non-ignored rows, optional label smoothing, like ``torch.nn.CrossEntropyLoss`` under bf16 autocast."""
import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr


import os
_CHECK_TARGETS = os.environ.get('TOK_CHECK_TARGETS', '0') == '1'


class _SoftmaxCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, ignore_index: int, smooth: float = 0.0):
        require_device(logits)
        ctx.spatial = None
        if logits.dim() == 4:
            # (N, C, H, W) logits vs (N, H, W) targets: every pixel is a row of the channel-last buffer
            n, classes, h, w = logits.shape
            zp = logits.detach().permute(0, 2, 3, 1)
            ld = zp.stride(2)
            if zp.dtype == BF16 and zp.stride(3) == 1 and ld >= classes and zp.stride(1) == w * ld \
                    and zp.stride(0) == h * w * ld:
                z = torch.as_strided(zp, (n * h * w, classes), (ld, 1))      # zero copy (segmentation head output)
            else:
                z = zp.to(BF16).contiguous().view(n * h * w, classes)
            ctx.spatial = (n, h, w)
            target = target.reshape(-1)
        else:
            z = logits.detach()
            if z.dtype != BF16 or z.stride(1) != 1:
                z = z.to(BF16).contiguous()
        rows, classes = z.shape
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.to(torch.int64).contiguous()
        if _CHECK_TARGETS:
            # torch raises "Target t is out of bounds" (a device-side assert on GPUs); the kernels drop such rows
            # consistently instead (no per-step host sync) — this opt-in check restores the raise for debugging
            bad = (target != ignore_index) & ((target < 0) | (target >= classes))
            if bool(bad.any()):
                raise IndexError(f'Target {int(target[bad][0])} is out of bounds.')
        dev = z.device
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty(_C.TOK_CE_LOSS_FLOATS, dtype=torch.float32, device=dev)
        _C.check(_C.lib().tok_softmax_ce_smooth_fwd(ptr(z), ptr(target), rows, classes, z.stride(0), ignore_index,
                                                    float(smooth), ptr(lse), ptr(row_loss), ptr(loss), stream_ptr()),
                 'tok_softmax_ce_smooth_fwd')
        ctx.z, ctx.target, ctx.lse, ctx.loss, ctx.ignore_index, ctx.smooth = z, target, lse, loss, ignore_index, float(smooth)
        ctx.in_dtype = logits.dtype
        n_valid = loss[1]             # number of non-ignored rows, stays on the device (reduction='sum')
        ctx.mark_non_differentiable(n_valid)
        return loss[0], n_valid

    @staticmethod
    def backward(ctx, g, _unused=None):
        z, target = ctx.z, ctx.target
        rows, classes = z.shape
        ld = z.stride(0)          # the kernel writes d(logits) in the row pitch of the logits
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty((rows, ld), dtype=BF16, device=z.device)
        _C.check(_C.lib().tok_softmax_ce_smooth_bwd(ptr(z), ptr(target), ptr(ctx.lse), ptr(ctx.loss), ptr(gs), rows,
                                                    classes, z.stride(0), ctx.ignore_index, ctx.smooth, ptr(d), stream_ptr()),
                 'tok_softmax_ce_smooth_bwd')
        ctx.z = ctx.target = ctx.lse = ctx.loss = None
        if ctx.spatial is not None:
            n, h, w = ctx.spatial
            if ld == pad8(classes) != classes:
                mark_padded(d)
            d = d.view(n, h, w, ld)[..., :classes].permute(0, 3, 1, 2)
        elif ld != classes:
            if ld == pad8(classes):
                mark_padded(d)
            d = d[:, :classes]
        if ctx.in_dtype != BF16:
            d = d.to(ctx.in_dtype)
        return d, None, None, None


@LOSSES.register_class
class CrossEntropyLoss(nn.Module):
    def __init__(self, weight=None, size_average=None, ignore_index: int = -100, reduce=None,
                 reduction: str = 'mean', label_smoothing: float = 0.0):
        super().__init__()
        if weight is not None or reduction not in ('mean', 'sum') or size_average is not None or reduce is not None:
            raise NotImplementedError("torchok_amd CrossEntropyLoss: reduction 'mean' or 'sum', no class weights")
        if not 0.0 <= label_smoothing <= 1.0:
            raise ValueError(f'label_smoothing must be between 0.0 and 1.0. Got: {label_smoothing}')
        self.ignore_index, self.label_smoothing, self.reduction = ignore_index, label_smoothing, reduction

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if input.dim() not in (2, 4):
            raise NotImplementedError('torchok_amd CrossEntropyLoss: (N, C) or (N, C, H, W) logits')
        if input.dim() == 4 and target.shape != (input.shape[0],) + tuple(input.shape[2:]):
            raise ValueError(f'Expected target of shape (N, H, W) for (N, C, H, W) logits, got {tuple(target.shape)}')
        loss, n_valid = _SoftmaxCE.apply(input, target, self.ignore_index, self.label_smoothing)
        # 'sum' = mean over the non-ignored rows * their number
        return loss if self.reduction == 'mean' else loss * n_valid
