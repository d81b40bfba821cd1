"""L1Loss / MSELoss / SmoothL1Loss / HuberLoss under the torch class names the reference registers
(``torchok/losses/__init__.py:13,19,23,24``): fp32 math on the bf16 prediction, fp64 two-stage fold, 'mean' or 'sum'."""
import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from .cross_entropy import materialized
from ..engine.core import BF16, ptr, require_device, stream_ptr


class _Regression(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input: Tensor, target: Tensor, kind: int, knee: float, mean: bool):
        require_device(input)
        x = input.detach().to(BF16).contiguous().view(-1)
        t = target.detach().to(torch.float32).contiguous().view(-1)
        loss = torch.empty(_C.TOK_CE_LOSS_FLOATS, dtype=torch.float32, device=x.device)
        _C.check(_C.lib().tok_regression_loss_fwd(ptr(x), ptr(t), x.numel(), kind, float(knee), int(mean), ptr(loss),
                                                  stream_ptr()), 'tok_regression_loss_fwd')
        ctx.saved = (x, t, kind, float(knee), int(mean), tuple(input.shape), input.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, t, kind, knee, mean, shape, dtype = ctx.saved
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        dx = torch.empty_like(x)
        _C.check(_C.lib().tok_regression_loss_bwd(ptr(x), ptr(t), ptr(gs), x.numel(), kind, knee, mean, ptr(dx),
                                                  stream_ptr()), 'tok_regression_loss_bwd')
        ctx.saved = None
        dx = dx.view(shape)
        return (dx if dtype == BF16 else dx.to(dtype)), None, None, None, None


class _RegressionLoss(nn.Module):
    kind = 0

    def __init__(self, size_average=None, reduce=None, reduction: str = 'mean', knee: float = 1.0):
        super().__init__()
        if size_average is not None or reduce is not None or reduction not in ('mean', 'sum'):
            raise NotImplementedError(f"torchok_amd {type(self).__name__}: reduction 'mean' or 'sum'")
        self.reduction, self.knee = reduction, knee

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        input = materialized(input)   # an UpsampledLogits would hide the autograd edge from Function.apply
        if tuple(input.shape) != tuple(target.shape):
            raise ValueError(f'{type(self).__name__}: input {tuple(input.shape)} and target {tuple(target.shape)} differ '
                             f'(broadcasting is not built)')
        return _Regression.apply(input, target, self.kind, self.knee, self.reduction == 'mean')


@LOSSES.register_class
class L1Loss(_RegressionLoss):
    kind = 0

    def __init__(self, size_average=None, reduce=None, reduction: str = 'mean'):
        super().__init__(size_average, reduce, reduction)


@LOSSES.register_class
class MSELoss(_RegressionLoss):
    kind = 1

    def __init__(self, size_average=None, reduce=None, reduction: str = 'mean'):
        super().__init__(size_average, reduce, reduction)


@LOSSES.register_class
class SmoothL1Loss(_RegressionLoss):
    kind = 2

    def __init__(self, size_average=None, reduce=None, reduction: str = 'mean', beta: float = 1.0):
        if beta == 0:           # torch: beta = 0 is exactly L1
            super().__init__(size_average, reduce, reduction)
            self.kind = 0
        else:
            super().__init__(size_average, reduce, reduction, knee=beta)
        self.beta = beta


@LOSSES.register_class
class HuberLoss(_RegressionLoss):
    kind = 3

    def __init__(self, reduction: str = 'mean', delta: float = 1.0):
        super().__init__(None, None, reduction, knee=delta)
        self.delta = delta
