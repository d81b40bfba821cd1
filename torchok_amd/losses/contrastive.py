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


@LOSSES.register_class
class ContrastiveLoss(nn.Module):
    def __init__(self, margin: float, reg: Optional[str] = None, reduction: Optional[str] = 'mean',
                 eps: Optional[float] = 1e-3):
        super().__init__()
        if reg is not None:
            if reg not in ('L1', 'L2'):
                raise ValueError(f'Unknown regularization type: {reg}')
            raise NotImplementedError('torchok_amd ContrastiveLoss: embedding regularisers are not built')
        if reduction != 'mean':
            if reduction != 'sum':
                raise ValueError(f'Unknown reduction type: {reduction}')
            raise NotImplementedError("torchok_amd ContrastiveLoss: reduction='mean' only")
        self.margin, self.reg, self.reduction, self.eps = margin, reg, reduction, eps

    def forward(self, emb1: Tensor, emb2: Tensor, R: Tensor) -> Tensor:
        return _Contrastive.apply(emb1, emb2, R, self.margin)
