"""Engine units of the object-contextual-representation head (reference ``torchok/models/heads/segmentation/ocr.py``).

spatial_gather     SpatialGather_Module.forward (:37-46): class context = softmax-over-pixels(aux logits)^T x features
object_attention   ObjectAttentionBlock.forward (:81-104, scale = 1): softmax_k(q key^T / sqrt(C)) value per pixel
channel_dropout    nn.Dropout2d of SpatialOCR (:121-124): one keep/scale factor per (image, channel)

Pixel tensors are NHWC bf16 TTensors; the per-class tensors ([B][K] rows of C channels: the "proxy" of the reference, a
(B, C, K, 1) map) are TTensors of shape (B, K, 1, Cp) so that the 1x1 ConvBnRelu units run on them unchanged; the
pixel-by-class weights (p, sim and their gradients) are fp32 [B][N][K] scratch tensors of the unit."""
from typing import Optional

import torch

from .. import _C
from .core import BF16, Node, Region, TTensor, grad_target, ptr, stream_ptr

F32 = torch.float32


def _geo(x: TTensor):
    if len(x.shape) == 2:             # (rows, channels) embeddings: one pixel per row
        return x.shape[0], 1, x.shape[1]
    n, h, w, cp = x.shape
    return n, h * w, cp


def _pool(lib, st, w, x: TTensor, k: int, scale: float, out: torch.Tensor, ldo: int, acc: int):
    b, npix, ldx = _geo(x)
    chunks = lib.tok_weighted_pool_chunks(npix)
    partial = torch.empty((b, chunks, k, x.c), dtype=F32, device=x.data.device)
    if not acc and out.shape[-1] != x.c:
        out[..., x.c:] = 0                       # the fold writes the logical channels only
    _C.check(lib.tok_weighted_pool(ptr(w), ptr(x.data), ldx, b, npix, k, x.c, scale, ptr(partial), ptr(out), ldo, acc, st),
             'tok_weighted_pool')
    return partial


class _GatherNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        feats, logits, p, k = self.feats, self.logits, self.p, self.k
        b, npix, ldf = _geo(feats)
        if feats.requires_grad:       # d feats[n] += sum_k p[n][k] dctx[k]
            tgt, acc = grad_target(feats)
            _C.check(lib.tok_class_pix_expand(ptr(p), ptr(g), g.shape[-1], b, npix, k, feats.c, 1.0, ptr(tgt), ldf, acc, st),
                     'tok_class_pix_expand')
        if logits.requires_grad:      # dp[n][k] = <feats[n], dctx[k]>, then through the softmax over pixels
            dp = torch.empty_like(p)
            _C.check(lib.tok_pix_class_matmul(ptr(feats.data), ldf, ptr(g), g.shape[-1], b, npix, k, feats.c, 1.0, ptr(dp), st),
                     'tok_pix_class_matmul')
            tgt, acc = grad_target(logits)
            _C.check(lib.tok_softmax_cols_bwd(ptr(p), ptr(dp), b, npix, k, self.scale, ptr(tgt), logits.cp, acc, st),
                     'tok_softmax_cols_bwd')
        self.out.grad = None

    def release(self):
        self.feats = self.logits = self.p = self.out = None


def spatial_gather(region: Region, feats: TTensor, logits: TTensor, scale: float = 1.0) -> TTensor:
    """(B, H, W, C) features x (B, H, W, K) auxiliary logits -> class context (B, K, 1, C)."""
    lib, st = _C.lib(), stream_ptr()
    b, npix, ldf = _geo(feats)
    k = logits.c
    if k > 64 or k * feats.c > 10240:
        raise NotImplementedError(f'spatial_gather: at most 64 classes and classes x channels <= 10240 (got {k} x {feats.c})')
    dev = feats.data.device
    p = torch.empty((b, npix, k), dtype=F32, device=dev)
    _C.check(lib.tok_softmax_cols_fwd(ptr(logits.data), logits.cp, b, npix, k, scale, ptr(p), st), 'tok_softmax_cols_fwd')
    out_data = torch.zeros((b, k, 1, feats.cp), dtype=BF16, device=dev)
    _pool(lib, st, p, feats, k, 1.0, out_data, feats.cp, 0)
    req = region.grad_mode and (feats.requires_grad or logits.requires_grad)
    out = TTensor(out_data, feats.c, requires_grad=req)
    if req:
        node = _GatherNode()
        node.feats, node.logits, node.p, node.k, node.scale, node.out = feats, logits, p, k, scale, out
        out.node = node
        for t in (feats, logits):
            if t.requires_grad:
                t.uses += 1
        region.add(node)
    return out


class _ObjAttnNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        q, key, value, sim, scale, k = self.q, self.key, self.value, self.sim, self.scale, self.k
        b, npix, ldq = _geo(q)
        c = q.c
        gt = TTensor(g, c)
        if value.requires_grad:       # dvalue[k] = sum_n sim[n][k] dctx[n]
            tgt, acc = grad_target(value)
            self.keep = [_pool(lib, st, sim, gt, k, 1.0, tgt, value.cp, acc)]
        if q.requires_grad or key.requires_grad:
            dsim = torch.empty_like(sim)
            _C.check(lib.tok_pix_class_matmul(ptr(g), g.shape[-1], ptr(value.data), value.cp, b, npix, k, c, 1.0, ptr(dsim), st),
                     'tok_pix_class_matmul')
            dlogit = torch.empty_like(sim)
            _C.check(lib.tok_softmax_rows_bwd_f32(ptr(sim), ptr(dsim), b * npix, k, ptr(dlogit), st), 'tok_softmax_rows_bwd_f32')
            if key.requires_grad:     # dkey[k] = scale * sum_n dlogit[n][k] q[n]
                tgt, acc = grad_target(key)
                self.keep = getattr(self, 'keep', []) + [_pool(lib, st, dlogit, q, k, scale, tgt, key.cp, acc)]
            if q.requires_grad:       # dq[n] = scale * sum_k dlogit[n][k] key[k]
                tgt, acc = grad_target(q)
                _C.check(lib.tok_class_pix_expand(ptr(dlogit), ptr(key.data), key.cp, b, npix, k, c, scale, ptr(tgt), ldq,
                                                  acc, st), 'tok_class_pix_expand')
        self.out.grad = None

    def release(self):
        self.q = self.key = self.value = self.sim = self.out = None
        self.keep = None


def object_attention(region: Region, q: TTensor, key: TTensor, value: TTensor, scale: float) -> TTensor:
    """q (B, H, W, C), key / value (B, K, 1, C) -> context (B, H, W, C)."""
    lib, st = _C.lib(), stream_ptr()
    b, npix, ldq = _geo(q)
    k, c = key.shape[1], q.c
    if key.shape[0] != b or value.shape[:2] != key.shape[:2] or key.c != c or value.c != c:
        raise ValueError('object_attention: key / value must be (B, K, 1, C) with the channels of q')
    if k > 64 or k * c > 10240:
        raise NotImplementedError(f'object_attention: at most 64 classes and classes x channels <= 10240 (got {k} x {c})')
    dev = q.data.device
    logit = torch.empty((b, npix, k), dtype=F32, device=dev)
    _C.check(lib.tok_pix_class_matmul(ptr(q.data), ldq, ptr(key.data), key.cp, b, npix, k, c, scale, ptr(logit), st),
             'tok_pix_class_matmul')
    sim = torch.empty_like(logit)
    _C.check(lib.tok_softmax_rows_f32(ptr(logit), b * npix, k, ptr(sim), st), 'tok_softmax_rows_f32')
    out_data = torch.empty_like(q.data)
    _C.check(lib.tok_class_pix_expand(ptr(sim), ptr(value.data), value.cp, b, npix, k, c, 1.0, ptr(out_data), ldq, 0, st),
             'tok_class_pix_expand')
    req = region.grad_mode and (q.requires_grad or key.requires_grad or value.requires_grad)
    out = TTensor(out_data, c, requires_grad=req)
    if req:
        node = _ObjAttnNode()
        node.q, node.key, node.value, node.sim, node.scale, node.k, node.out = q, key, value, sim, scale, k, out
        out.node = node
        for t in (q, key, value):
            if t.requires_grad:
                t.uses += 1
        region.add(node)
    return out


class _ChannelScaleNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        b, npix, ld = _geo(self.x)
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_channel_scale(ptr(g), ptr(self.s), ptr(tgt), acc, b, npix, self.x.c, ld, stream_ptr()),
                 'tok_channel_scale')
        self.out.grad = None

    def release(self):
        self.x = self.s = self.out = None


def channel_dropout(region: Region, x: TTensor, scale: Optional[torch.Tensor]) -> TTensor:
    """x * scale[image][channel] (scale = keep mask / (1 - p), fp32 (B, C)); None = identity (eval mode)."""
    if scale is None:
        return x
    b, npix, ld = _geo(x)
    y = torch.empty_like(x.data)
    s = scale.to(F32).contiguous()
    _C.check(_C.lib().tok_channel_scale(ptr(x.data), ptr(s), ptr(y), 0, b, npix, x.c, ld, stream_ptr()), 'tok_channel_scale')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _ChannelScaleNode()
        node.x, node.s, node.out = x, s, out
        out.node = node
        x.uses += 1
        region.add(node)
    return out
