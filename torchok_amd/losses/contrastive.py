"""ContrastiveLoss on the fused pairwise-distance kernels (reference
``torchok/losses/representation/pairwise.py:8-136``: BasePairwiseLoss -> GeneralPairWeightingLoss ->
ContrastiveLoss; called as ``loss(emb1=..., emb2=..., R=...)`` by JointLoss)."""
from typing import Optional

import torch
from torch import Tensor, nn

from .. import _C
from ..constructor import LOSSES
from ..engine.core import BF16, mark_padded, pad8, ptr, require_device, stream_ptr


class _Contrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb1: Tensor, emb2: Tensor, R: Tensor, margin: float):
        require_device(emb1)
        same = emb1.data_ptr() == emb2.data_ptr() and emb1.shape == emb2.shape and emb1.stride() == emb2.stride()

        def prep(e):
            e = e.detach()
            if e.dtype != BF16 or e.stride(1) != 1:
                e = e.to(BF16).contiguous()
            return e
        e1 = prep(emb1)
        e2 = e1 if same else prep(emb2)
        n1, d = e1.shape
        n2 = e2.shape[0]
        if e1.stride(0) != e2.stride(0):
            e2 = e2.contiguous()
            e1 = e1.contiguous()
        r = R.detach().to(torch.float32).contiguous()
        dev = e1.device
        S = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        row_loss = torch.empty(n1, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        _C.check(_C.lib().tok_contrastive_fwd(ptr(e1), ptr(e2), ptr(r), n1, n2, d, e1.stride(0), float(margin), ptr(S),
                                              ptr(row_loss), ptr(loss), stream_ptr()), 'tok_contrastive_fwd')
        ctx.saved = (e1, e2, r, S, same, float(margin), emb1.dtype, emb2.dtype)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        e1, e2, r, S, same, margin, dt1, dt2 = ctx.saved
        n1, d = e1.shape
        n2 = e2.shape[0]
        ld = pad8(d)
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        # gradients are produced in zero-padded rows so the producing head region can take them as is
        de1 = torch.zeros((n1, ld), dtype=BF16, device=e1.device)
        de2 = None if same else torch.zeros((n2, ld), dtype=BF16, device=e1.device)
        if e1.stride(0) != ld:
            e1c = torch.zeros((n1, ld), dtype=BF16, device=e1.device)
            e1c[:, :d] = e1
            e2c = e1c if same else torch.zeros((n2, ld), dtype=BF16, device=e1.device)
            if not same:
                e2c[:, :d] = e2
        else:
            e1c, e2c = e1, e2
        _C.check(_C.lib().tok_contrastive_bwd(ptr(e1c), ptr(e2c), ptr(r), ptr(S), ptr(gs), n1, n2, d, ld, margin,
                                              ptr(de1), ptr(de2), int(same), stream_ptr()), 'tok_contrastive_bwd')
        ctx.saved = None

        def fin(t, dt):
            if ld != d:
                mark_padded(t)
                t = t[:, :d]
            return t if dt == BF16 else t.to(dt)
        if same:
            # emb1 and emb2 are one tensor (pairwise_task.py:79): autograd adds the two returned grads,
            # so the whole gradient rides on the first and the second is None (= zero)
            return fin(de1, dt1), None, None, None
        return fin(de1, dt1), fin(de2, dt2), None, None


class _EmbedReg(torch.autograd.Function):
    """mean_i reg_i of BasePairwiseLoss.regularize (pairwise.py:28-46): 'L1' sum|e_i| or 'L2' ||e_i||_2."""

    @staticmethod
    def forward(ctx, emb: Tensor, mode: int):
        require_device(emb)
        e = emb.detach()
        if e.dtype != BF16 or e.stride(1) != 1:
            e = e.to(BF16).contiguous()
        n, d = e.shape
        row_reg = torch.empty(n, dtype=torch.float32, device=e.device)
        out = torch.empty(1, dtype=torch.float32, device=e.device)
        _C.check(_C.lib().tok_embed_reg_fwd(ptr(e), n, d, e.stride(0), mode, ptr(row_reg), ptr(out), stream_ptr()),
                 'tok_embed_reg_fwd')
        ctx.saved = (e, row_reg, mode, emb.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        e, row_reg, mode, dt = ctx.saved
        n, d = e.shape
        ld = pad8(d)
        if e.stride(0) != ld:
            ec = torch.zeros((n, ld), dtype=BF16, device=e.device)
            ec[:, :d] = e
        else:
            ec = e
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        de = torch.empty((n, ld), dtype=BF16, device=e.device)
        _C.check(_C.lib().tok_embed_reg_bwd(ptr(ec), ptr(row_reg), ptr(gs), 1.0 / n, n, d, ld, mode, ptr(de), stream_ptr()),
                 'tok_embed_reg_bwd')
        ctx.saved = None
        if ld != d:
            mark_padded(de)
            de = de[:, :d]
        return (de if dt == BF16 else de.to(dt)), None


@LOSSES.register_class
class ContrastiveLoss(nn.Module):
    def __init__(self, margin: float, reg: Optional[str] = None, reduction: Optional[str] = 'mean',
                 eps: Optional[float] = 1e-3):
        super().__init__()
        if reg not in (None, 'L1', 'L2'):
            raise ValueError(f'Unknown regularization type: {reg}')
        if reduction not in ('mean', 'sum'):
            raise ValueError(f'Unknown reduction type: {reduction}')
        self.margin, self.reg, self.reduction, self.eps = margin, reg, reduction, eps

    def forward(self, emb1: Tensor, emb2: Tensor, R: Tensor) -> Tensor:
        # mean over the rows of (pair term + eps * regulariser of emb1), pairwise.py:101-103; 'sum' = mean * rows
        loss = _Contrastive.apply(emb1, emb2, R, self.margin)
        if self.reg is not None:
            loss = loss + self.eps * _EmbedReg.apply(emb1, 1 if self.reg == 'L1' else 2)
        if self.reduction == 'sum':
            loss = loss * emb1.shape[0]
        return loss
