"""Engine units of the multi-resolution rows (HRNet backbone / segmentation neck / segmentation head).

Unit                 replaces
-------------------  --------------------------------------------------------------------------------
fuse_sum_relu        [timm] HighResolutionModule.forward: relu(sum_j fuse_ij(x_j)), with the nearest
                     nn.Upsample of the low-resolution terms folded into the summation kernel
bilinear_concat      F.interpolate(bilinear, align_corners=False) x3 + torch.cat
                     (necks/segmentation/hrnet.py:36-41): every source is written straight into its
                     channel slice of the concat buffer
bilinear_resize      F.interpolate(segm_logits, size=input.shape[2:], mode='bilinear')
                     (heads/segmentation/base.py:37)
"""
from typing import List, Sequence, Tuple

import torch

from .. import _C
from .core import BF16, Node, Region, TTensor, await_ready, grad_target, pad8, ptr, stream_ptr


class _FuseSumNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None:
            return
        n, h, w, c = self.out.shape
        lib, st = _C.lib(), stream_ptr()
        for t, sh in self.terms:
            if not t.requires_grad:
                continue
            tgt, acc = grad_target(t)
            _C.check(lib.tok_fuse_sum_relu_bwd(ptr(g), ptr(self.mask), n, h, w, c, sh, ptr(tgt), acc, st),
                     'tok_fuse_sum_relu_bwd')
        self.out.grad = None

    def release(self):
        self.terms = self.out = self.mask = None


def fuse_sum_relu(region: Region, terms: Sequence[Tuple[TTensor, int]], relu: bool = True) -> TTensor:
    """out = relu(sum_j upsample_nearest(t_j, 2**shift_j)); the first term fixes the output shape (shift 0)."""
    if not 1 <= len(terms) <= 4:
        raise NotImplementedError('fuse_sum_relu: 1..4 terms')
    await_ready(*(t for t, _ in terms))
    ref, sh0 = terms[0]
    n, h, w, cp = ref.shape
    h, w = h << sh0, w << sh0
    for t, sh in terms:
        tn, th, tw, tc = t.shape
        if (tn, th << sh, tw << sh, tc) != (n, h, w, cp) or t.c != ref.c:
            # the reference fails here as well: nn.Upsample(2^k) of a ceil-halved map does not match (y + ...)
            raise ValueError(f'HRNet fuse: branch map {t.shape} x{1 << sh} does not match {(n, h, w, cp)}; '
                             f'input height/width must be divisible by 32')
    # a term that carries `.affine` is the raw output of a unit without activation: its BatchNorm apply is one fma here
    affine = any(t.affine is not None for t, _ in terms)
    args = []
    for i in range(4):
        if i < len(terms):
            t = terms[i][0]
            args += [ptr(t.data), terms[i][1]]
            if affine:
                args += [ptr(t.affine[0]), ptr(t.affine[1])] if t.affine is not None else [None, None]
        else:
            args += [None, 0] + ([None, None] if affine else [])
    dev = ref.data.device
    out_data = torch.empty((n, h, w, cp), dtype=BF16, device=dev)
    req = region.grad_mode and any(t.requires_grad for t, _ in terms)
    mask = torch.empty((n * h * w, cp // 8), dtype=torch.uint8, device=dev) if (req and relu) else None
    if affine:
        _C.check(_C.lib().tok_fuse_sum_affine_relu_fwd(*args, n, h, w, cp, int(relu), ptr(out_data), ptr(mask), stream_ptr()),
                 'tok_fuse_sum_affine_relu_fwd')
    else:
        _C.check(_C.lib().tok_fuse_sum_relu_fwd(*args, n, h, w, cp, int(relu), ptr(out_data), ptr(mask), stream_ptr()),
                 'tok_fuse_sum_relu_fwd')
    out = TTensor(out_data, ref.c, requires_grad=req)
    if req:
        node = _FuseSumNode()
        node.terms, node.out, node.mask = list(terms), out, mask
        out.node = node
        for t, _ in terms:
            if t.requires_grad:
                t.uses += 1
        region.add(node)
    return out


class _BilinearNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None:
            return
        n, hd, wd, ld = self.out.shape
        lib, st = _C.lib(), stream_ptr()
        off = 0
        for t, cw in zip(self.srcs, self.widths):
            _, hs, ws, cp = t.shape
            if t.requires_grad:
                tgt, acc = grad_target(t)
                if cw != cp and not acc:
                    tgt.zero_()           # padding channels of a fresh gradient buffer
                _C.check(lib.tok_bilinear_bwd(ptr(g), n, hd, wd, ld, off, ptr(tgt), hs, ws, cw, cp, acc, st),
                         'tok_bilinear_bwd')
            off += cw
        self.out.grad = None

    def release(self):
        self.srcs = self.out = self.widths = None


def bilinear_concat(region: Region, srcs: List[TTensor], size: Tuple[int, int]) -> TTensor:
    """cat([interpolate(s, size, 'bilinear', align_corners=False) for s in srcs], dim=channel)."""
    await_ready(*srcs)
    n = srcs[0].shape[0]
    hd, wd = int(size[0]), int(size[1])
    # a single map keeps its padded width; concatenated maps are packed at their LOGICAL channel offsets
    # (18 + 36 + 72 + 144 = 270 -> row pitch 272), which costs the 16-byte fast path only for odd widths
    widths = [srcs[0].cp] if len(srcs) == 1 else [t.c for t in srcs]
    ld = pad8(sum(widths))
    dev = srcs[0].data.device
    alloc = torch.zeros if ld != sum(widths) else torch.empty
    out_data = alloc((n, hd, wd, ld), dtype=BF16, device=dev)
    lib, st = _C.lib(), stream_ptr()
    off = 0
    for t, cw in zip(srcs, widths):
        _, hs, ws, cp = t.shape
        _C.check(lib.tok_bilinear_fwd(ptr(t.data), n, hs, ws, cw, cp, ptr(out_data), hd, wd, ld, off, st),
                 'tok_bilinear_fwd')
        off += cw
    req = region.grad_mode and any(t.requires_grad for t in srcs)
    out = TTensor(out_data, sum(t.c for t in srcs), requires_grad=req)
    if req:
        node = _BilinearNode()
        node.srcs, node.out, node.widths = list(srcs), out, widths
        out.node = node
        for t in srcs:
            if t.requires_grad:
                t.uses += 1
        region.add(node)
    return out


def bilinear_resize(region: Region, x: TTensor, size: Tuple[int, int]) -> TTensor:
    return bilinear_concat(region, [x], size)
