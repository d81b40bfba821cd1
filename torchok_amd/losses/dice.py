"""DiceLoss on the fused Dice kernels (reference ``torchok/losses/segmentation/dice.py:86-188``; used next to
CrossEntropyLoss by the shipped HRNet recipe, ``examples/configs/segmentation_sweet_pepper.yaml:16-27``).
'multiclass' (softmax + one-hot), 'binary' (sigmoid) and 'multilabel' (sigmoid per class) modes, from logits."""
from typing import List

import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from .cross_entropy import materialized
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr

BINARY_MODE, MULTICLASS_MODE, MULTILABEL_MODE = 'binary', 'multiclass', 'multilabel'


def _pixel_rows(logits: Tensor, classes: int):
    """(N, C, H, W) logits -> bf16 [N*H*W][ld] rows (zero copy for the channel-last output of SegmentationHead)."""
    n, c, h, w = logits.shape
    zp = logits.detach().permute(0, 2, 3, 1)
    ld = zp.stride(2)
    if zp.dtype == BF16 and zp.stride(3) == 1 and classes <= ld <= 64 and zp.stride(1) == w * ld \
            and zp.stride(0) == h * w * ld:
        return torch.as_strided(zp, (n * h * w, classes), (ld, 1)), ld
    ld = pad8(classes)
    z = torch.zeros((n * h * w, ld), dtype=BF16, device=logits.device)
    z[:, :classes] = zp.reshape(n * h * w, c)[:, :classes]
    return z[:, :classes], ld


class _Dice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, mode: int, smooth: float, eps: float, log_loss: bool, sel):
        require_device(logits)
        if mode == 0:
            n, classes, h, w = logits.shape
            z, ld = _pixel_rows(logits, classes)
            tgt = target.reshape(-1).to(torch.int64).contiguous()
            ctx.view = (n, h, w, classes)
        elif mode == 2:   # multilabel: (N, C, H, W) logits and targets; the targets follow the logits into pixel rows
            n, classes, h, w = logits.shape
            z, ld = _pixel_rows(logits, classes)
            tgt = target.detach().permute(0, 2, 3, 1).to(torch.float32).contiguous().view(n * h * w, classes)
            ctx.view = (n, h, w, classes)
        else:   # binary: (N, H, W) logits — column 0 of the one-class head output, or any tensor
            classes = 1
            flat = logits.detach()
            if flat.dtype == BF16 and flat.dim() == 3 and flat.stride(2) in (8,) and flat.stride(1) == flat.shape[2] * 8 \
                    and flat.stride(0) == flat.shape[1] * flat.shape[2] * 8:
                ld = 8
                z = torch.as_strided(flat, (flat.numel(), 1), (8, 1))
            else:
                ld = 8
                z = torch.zeros((flat.numel(), 8), dtype=BF16, device=logits.device)
                z[:, 0] = flat.reshape(-1)
                z = z[:, :1]
            tgt = target.reshape(-1).to(torch.float32).contiguous()
            ctx.view = tuple(logits.shape)
        rows = z.shape[0]
        dev = z.device
        lib = _C.lib()
        partial = torch.empty((lib.tok_dice_rows(rows), 3, classes), dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        coef = torch.empty((2, classes), dtype=torch.float32, device=dev)
        _C.check(lib.tok_dice_fwd(ptr(z), ptr(tgt), rows, classes, ld, mode, float(smooth), float(eps), int(log_loss),
                                  ptr(sel), 0 if sel is None else sel.numel(), ptr(partial), ptr(loss), ptr(coef),
                                  stream_ptr()), 'tok_dice_fwd')
        ctx.saved = (z, tgt, coef, ld, mode, classes, logits.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        z, tgt, coef, ld, mode, classes, in_dtype = ctx.saved
        rows = z.shape[0]
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty((rows, ld), dtype=BF16, device=z.device)
        _C.check(_C.lib().tok_dice_bwd(ptr(z), ptr(tgt), ptr(coef), ptr(gs), rows, classes, ld, mode, ptr(d), stream_ptr()),
                 'tok_dice_bwd')
        ctx.saved = None
        if mode in (0, 2):
            n, h, w, c = ctx.view
            if ld == pad8(c) != c:
                mark_padded(d)
            d = d.view(n, h, w, ld)[..., :c].permute(0, 3, 1, 2)
        else:
            d = d[:, 0].reshape(ctx.view)
        if in_dtype != BF16:
            d = d.to(in_dtype)
        return d, None, None, None, None, None, None


@LOSSES.register_class
class DiceLoss(nn.Module):
    def __init__(self, mode: str, classes: List[int] = None, log_loss: bool = False, from_logits: bool = True,
                 smooth: float = 0, eps: float = 1e-7):
        if mode not in {BINARY_MODE, MULTILABEL_MODE, MULTICLASS_MODE}:
            raise ValueError(f'DiceLoss initialize. Mode {mode} does not supper. Please choose one of from'
                             f'{[BINARY_MODE, MULTILABEL_MODE, MULTICLASS_MODE]}.')
        super().__init__()
        if classes is not None and mode == BINARY_MODE:
            raise ValueError('DiceLoss initialize. Masking classes is not supported with mode=binary')
        if not from_logits:
            raise NotImplementedError('torchok_amd DiceLoss: from logits only (the activation is fused into the kernels)')
        self.mode = mode
        self.classes = None if classes is None else torch.as_tensor(list(classes), dtype=torch.long)
        self.from_logits, self.smooth, self.eps, self.log_loss = from_logits, smooth, eps, log_loss

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        input = materialized(input)   # an UpsampledLogits would hide the autograd edge from Function.apply
        if self.mode in (BINARY_MODE, MULTILABEL_MODE) and input.shape != target.shape:
            raise ValueError(f"Shapes of input {input.shape} and target {target.shape} tensors don't match!")
        if self.mode == MULTICLASS_MODE and input[:, 0].shape != target.shape:
            raise ValueError(f"Shapes of input {input.shape} and target {target.shape} tensors don't match!")
        sel = None
        if self.classes is not None:
            sel = self.classes.to(input.device)
        if self.mode == MULTILABEL_MODE and input.dim() != 4:
            raise NotImplementedError("torchok_amd DiceLoss 'multilabel': (N, C, H, W) logits")
        mode = {MULTICLASS_MODE: 0, BINARY_MODE: 1, MULTILABEL_MODE: 2}[self.mode]
        return _Dice.apply(input, target, mode, self.smooth, self.eps, self.log_loss, sel)
