"""Engine units of the metric-learning heads (ArcFace, normalised embeddings).

Unit                replaces
------------------  --------------------------------------------------------------------------
l2_normalize        F.normalize(x, p=2, dim=-1)            linear_head.py:33-34, arcface_head.py:125
cosine_linear       F.linear(x, F.normalize(W))            arcface_head.py:126-127
arcface_margin      ArcFaceHead.__add_margin               arcface_head.py:95-108
"""
import math

import torch
from torch import nn

from .. import _C
from .core import (BF16, Node, Region, TTensor, commit_param_grad, grad_target, pad8, param_grad_target, ptr,
                   stream_ptr)

F32 = torch.float32
EPS = 1e-12   # F.normalize default


class _L2NormNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        n, cp = self.x.shape
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_l2norm_bwd(ptr(g), ptr(self.out.data), ptr(self.inv), ptr(tgt), acc, n, self.x.c, cp,
                                         0, stream_ptr()), 'tok_l2norm_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = self.inv = None


def l2_normalize(region: Region, x: TTensor) -> TTensor:
    n, cp = x.shape
    y = torch.empty_like(x.data)
    inv = torch.empty(n, dtype=F32, device=x.data.device)
    _C.check(_C.lib().tok_l2norm_fwd(ptr(x.data), ptr(y), ptr(inv), n, x.c, cp, 0, EPS, stream_ptr()),
             'tok_l2norm_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _L2NormNode()
        node.x, node.out, node.inv = x, out, inv
        out.node = node
        x.uses += 1
        region.add(node)
    return out


class _CosineLinearNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        x, w, d = self.x, self.weight, self.desc
        k, c = w.shape
        if w.requires_grad:
            ws_bytes = lib.tok_conv_wgrad_ws_bytes(d)
            ws = torch.empty(max(ws_bytes // 4, 1), dtype=F32, device=g.device)
            dwhat = torch.empty((k, c), dtype=F32, device=g.device)
            _C.check(lib.tok_conv_wgrad(d, ptr(x.data), ptr(g), ptr(dwhat), k, c, ptr(ws), ws_bytes, 0, st),
                     'tok_conv_wgrad')
            slot, mode = param_grad_target(w)
            _C.check(lib.tok_l2norm_bwd(ptr(dwhat), ptr(self.what), ptr(self.winv), ptr(slot), 1 if mode == 1 else 0,
                                        k, c, c, 1, st), 'tok_l2norm_bwd')
            commit_param_grad(w, slot, mode)
        if x.requires_grad:
            tgt, acc = grad_target(x)
            _C.check(lib.tok_conv_dgrad(d, ptr(g), ptr(self.w_dgrad), ptr(tgt), acc, st), 'tok_conv_dgrad')
        self.out.grad = None

    def release(self):
        self.x = self.out = self.what = self.winv = self.w_dgrad = None


def cosine_linear(region: Region, x: TTensor, weight: nn.Parameter) -> TTensor:
    """out = x @ normalize(weight, dim=1)^T   (x is expected to be L2-normalised already)."""
    lib, st = _C.lib(), stream_ptr()
    k, c = weight.shape
    n, cp = x.shape
    if cp != c:
        raise NotImplementedError('cosine_linear: embedding width must be a multiple of 8')
    kp = pad8(k)
    dev = x.data.device
    what = torch.empty((k, c), dtype=F32, device=dev)
    winv = torch.empty(k, dtype=F32, device=dev)
    _C.check(lib.tok_l2norm_fwd(ptr(weight), ptr(what), ptr(winv), k, c, c, 1, EPS, st), 'tok_l2norm_fwd')
    need_dx = region.grad_mode and x.requires_grad
    w_fwd = torch.empty((kp, 1, 1, cp), dtype=BF16, device=dev)
    w_dgrad = torch.empty((cp, 1, 1, kp), dtype=BF16, device=dev) if need_dx else None
    if need_dx:
        _C.check(lib.tok_pack_weight_both(ptr(what), k, 1, 1, c, ptr(w_fwd), kp, 1, cp, ptr(w_dgrad), st),
                 'tok_pack_weight_both')
    else:
        _C.check(lib.tok_pack_weight_fwd(ptr(what), k, 1, 1, c, ptr(w_fwd), kp, 1, cp, st), 'tok_pack_weight_fwd')
    d = _C.ConvDesc(n, 1, 1, cp, kp, 1, 1, 1, 1, 1, 0, 1)
    y = torch.empty((n, kp), dtype=BF16, device=dev)
    _C.check(lib.tok_conv_fwd(d, ptr(x.data), ptr(w_fwd), None, ptr(y), None, st), 'tok_conv_fwd')
    req = region.grad_mode and (x.requires_grad or weight.requires_grad)
    out = TTensor(y, k, requires_grad=req)
    if req:
        node = _CosineLinearNode()
        node.x, node.out, node.weight, node.desc = x, out, weight, d
        node.what, node.winv, node.w_dgrad = what, winv, w_dgrad
        out.node = node
        if x.requires_grad:
            x.uses += 1
        region.add(node)
    return out


class _ArcMarginNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.cos.requires_grad:
            return
        n, ld = self.cos.shape
        tgt, acc = grad_target(self.cos)
        if acc:
            raise RuntimeError('arcface_margin: the cosine matrix has a single consumer')
        _C.check(_C.lib().tok_arcface_margin_bwd(ptr(self.cos.data), ptr(self.target), ptr(g), n, self.cos.c, ld,
                                                 *self.consts, ptr(tgt), stream_ptr()), 'tok_arcface_margin_bwd')
        self.out.grad = None

    def release(self):
        self.cos = self.out = self.target = None


def arcface_margin(region: Region, cos: TTensor, target: torch.Tensor, margin: float, scale: float,
                   easy_margin: bool) -> TTensor:
    n, ld = cos.shape
    if target.dtype != torch.int64 or not target.is_contiguous():
        target = target.to(torch.int64).contiguous()
    consts = (math.cos(margin), math.sin(margin), math.cos(math.pi - margin), math.sin(math.pi - margin) * margin,
              int(easy_margin), float(scale))
    y = torch.empty_like(cos.data)
    _C.check(_C.lib().tok_arcface_margin_fwd(ptr(cos.data), ptr(target), n, cos.c, ld, *consts, ptr(y), stream_ptr()),
             'tok_arcface_margin_fwd')
    req = region.grad_mode and cos.requires_grad
    out = TTensor(y, cos.c, requires_grad=req)
    if req:
        node = _ArcMarginNode()
        node.cos, node.out, node.target, node.consts = cos, out, target, consts
        out.node = node
        cos.uses += 1
        region.add(node)
    return out
