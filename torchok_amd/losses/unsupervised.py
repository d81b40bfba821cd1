"""NT_XentLoss (reference ``torchok/losses/representation/unsupervised.py:7-54``) and torch's ``TripletMarginLoss``
(registered by the reference at ``losses/__init__.py:39``; ``examples/configs/triplet_sop.yaml:20``) on fused kernels."""
import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr


def _rows_bf16(t: Tensor, ld: int) -> Tensor:
    """(n, d) embeddings -> zero-padded bf16 rows of pitch ld (zero copy when they already are)."""
    t = t.detach()
    n, d = t.shape
    if t.dtype == BF16 and t.stride(1) == 1 and t.stride(0) == ld:
        return t
    out = torch.zeros((n, ld), dtype=BF16, device=t.device)
    out[:, :d] = t
    return out[:, :d] if ld != d else out


def _grad_out(buf: Tensor, d: int, dtype):
    if buf.shape[1] != d:
        mark_padded(buf)
        buf = buf[:, :d]
    return buf if dtype == BF16 else buf.to(dtype)


class _NTXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb1: Tensor, emb2: Tensor, temperature: float):
        require_device(emb1)
        b, d = emb1.shape
        ld = pad8(d)
        e = torch.zeros((2 * b, ld), dtype=BF16, device=emb1.device)
        e[:b, :d] = emb1.detach()
        e[b:, :d] = emb2.detach()
        n = 2 * b
        lse = torch.empty(n, dtype=torch.float32, device=e.device)
        row_loss = torch.empty(n, dtype=torch.float32, device=e.device)
        loss = torch.empty(1, dtype=torch.float32, device=e.device)
        _C.check(_C.lib().tok_ntxent_fwd(ptr(e), n, d, ld, float(temperature), ptr(lse), ptr(row_loss), ptr(loss),
                                         stream_ptr()), 'tok_ntxent_fwd')
        ctx.saved = (e, lse, b, d, ld, float(temperature), emb1.dtype, emb2.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        e, lse, b, d, ld, temp, dt1, dt2 = ctx.saved
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        de = torch.empty_like(e)
        _C.check(_C.lib().tok_ntxent_bwd(ptr(e), ptr(lse), ptr(gs), 2 * b, d, ld, temp, ptr(de), stream_ptr()),
                 'tok_ntxent_bwd')
        ctx.saved = None
        d1, d2 = de[:b], de[b:]
        if ld != d:
            d1, d2 = d1[:, :d], d2[:, :d]
        return (d1 if dt1 == BF16 else d1.to(dt1)), (d2 if dt2 == BF16 else d2.to(dt2)), None


@LOSSES.register_class
class NT_XentLoss(nn.Module):
    def __init__(self, reduction: str = 'mean', temperature: float = 1.0) -> None:
        super().__init__()
        if reduction not in ('mean', 'sum'):
            raise NotImplementedError("torchok_amd NT_XentLoss: reduction 'mean' or 'sum'")
        self.reduction = reduction
        self.temperature = temperature

    def forward(self, emb1, emb2, emb_m=None):
        if emb_m is not None:
            raise NotImplementedError('torchok_amd NT_XentLoss: the memory-bank form (emb_m) is not built')
        if emb1.shape != emb2.shape or emb1.dim() != 2:
            raise ValueError(f'NT_XentLoss expects two (B, D) embeddings, got {tuple(emb1.shape)} and {tuple(emb2.shape)}')
        loss = _NTXent.apply(emb1, emb2, self.temperature)
        return loss if self.reduction == 'mean' else loss * (2 * emb1.shape[0])    # CrossEntropyLoss over the 2B rows


class _Triplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor: Tensor, positive: Tensor, negative: Tensor, margin: float, eps: float, swap: bool):
        require_device(anchor)
        n, d = anchor.shape
        ld = pad8(d)
        a, p, ng = (_rows_bf16(t, ld) for t in (anchor, positive, negative))
        dev = anchor.device
        dist = torch.empty((n, 3), dtype=torch.float32, device=dev)
        row_loss = torch.empty(n, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _C.check(_C.lib().tok_triplet_fwd(ptr(a), ptr(p), ptr(ng), n, d, ld, float(margin), float(eps), int(swap), ptr(dist),
                                          ptr(row_loss), ptr(loss), stream_ptr()), 'tok_triplet_fwd')
        ctx.saved = (a, p, ng, dist, n, d, ld, float(margin), float(eps), int(swap),
                     (anchor.dtype, positive.dtype, negative.dtype))
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        a, p, ng, dist, n, d, ld, margin, eps, swap, dts = ctx.saved
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        outs = [torch.empty((n, ld), dtype=BF16, device=a.device) for _ in range(3)]
        _C.check(_C.lib().tok_triplet_bwd(ptr(a), ptr(p), ptr(ng), ptr(dist), ptr(gs), n, d, ld, margin, eps, swap,
                                          ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), stream_ptr()), 'tok_triplet_bwd')
        ctx.saved = None
        return (*(_grad_out(o, d, dt) for o, dt in zip(outs, dts)), None, None, None)


@LOSSES.register_class
class TripletMarginLoss(nn.Module):
    """torch.nn.TripletMarginLoss semantics (p = 2, mean or sum reduction)."""

    def __init__(self, margin: float = 1.0, p: float = 2.0, eps: float = 1e-6, swap: bool = False, size_average=None,
                 reduce=None, reduction: str = 'mean'):
        super().__init__()
        if p != 2 or reduction not in ('mean', 'sum') or size_average is not None or reduce is not None:
            raise NotImplementedError("torchok_amd TripletMarginLoss: p=2, reduction 'mean' or 'sum'")
        self.margin, self.p, self.eps, self.swap, self.reduction = margin, p, eps, swap, reduction

    def forward(self, anchor: Tensor, positive: Tensor, negative: Tensor) -> Tensor:
        loss = _Triplet.apply(anchor, positive, negative, self.margin, self.eps, self.swap)
        return loss if self.reduction == 'mean' else loss * anchor.shape[0]
