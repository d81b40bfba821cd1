"""The HRNet segmentation neck as ONE engine unit, computed in the commuted order.

HRNetSegmentationNeck (necks/segmentation/hrnet.py:36-45) is

    feats = cat([x0] + [F.interpolate(x_j, size=x0.shape[2:], mode='bilinear') for j in 1..3], dim=1)
    out   = relu(bn(conv1x1(feats)))

A 1x1 convolution is a linear map across the channels of ONE pixel, the interpolation a linear map across the pixels of ONE
channel: they commute.  With W_j = the filter columns that belong to source j,

    conv1x1(cat_j up(x_j)) = sum_j up(conv1x1(x_j; W_j))

so the 720 x 720 product runs at every source's own resolution (48 + 96/4 + 192/16 + 384/64 = 90 input channels' worth of
MACs per output pixel instead of 720) and `tok_bilinear_sum_stats` adds the four results up and leaves the BatchNorm
statistics of the sum.  The 720-channel concat tensor (1.13 GB at 512x1024 B=24), its gradient, the 815-GFLOP product over it
and the 815-GFLOP weight gradient are never built: forward 0.47 (interpolate) + 1.31 (GEMM) ms -> four small GEMMs + one
streaming pass; backward GEMM 1.22 + weight gradient 2.73 + interpolate-transpose 1.47 ms -> three transposes of d(y), four
small data / weight gradients.  Same parameters, same state_dict, same result up to bf16 rounding (each of the four partial
products is rounded to bf16 before the sum; the direct order rounds the four interpolated inputs instead).
TOK_NECK_COMMUTE=0 keeps the direct order (bilinear_concat + conv_bn_act).
"""
import os
from typing import List, Optional, Tuple

import torch
from torch import nn

from .. import _C
from . import functional as EF
from . import resample as ER
from .core import (BF16, Region, TTensor, await_ready, commit_param_grad, grad_target, pad8, param_grad_target, ptr,
                   stream_ptr)

F32 = torch.float32
NECK_COMMUTE = os.environ.get('TOK_NECK_COMMUTE', '1') != '0'
NECK_WGRAD_SIDE = os.environ.get('TOK_NECK_WGRAD_SIDE', '1') != '0'


class _CommutedNeckNode(EF._ConvBnActNode):
    """relu(bn(sum_j up(conv1x1(x_j; W_j)))).  A _ConvBnActNode as far as BatchNorm goes (`y` = the summed pre-normalisation
    map, so a consumer's data gradient may fold this unit's BatchNorm-backward sums); its own convolution part below."""

    def release(self):
        super().release()
        self.srcs = self.descs = self.offs = None

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        out: TTensor = self.out
        g = out.grad
        if g is None:
            return
        conv, bn = self.conv, self.bn
        kp = self.y.shape[-1]
        m = self.y.numel() // kp
        n, h, w, _ = self.y.shape
        w_need = conv.weight.requires_grad
        srcs: List[TTensor] = self.srcs
        mask = self.mask if self.relu else None
        if self.fused_coef is not None:
            coef = self.fused_coef
        else:
            self._finalize_bwd(lib, st, g, mask, m, kp, bn.weight.requires_grad, bn.bias.requires_grad)
            coef = self.coef
        if not (w_need or any(t.requires_grad for t in srcs)):
            out.grad = None
            return
        dy = torch.empty_like(self.y)
        _C.check(lib.tok_bn_bwd_apply(ptr(g), ptr(self.y), ptr(mask), ptr(self.scale), ptr(self.shift), ptr(coef),
                                      int(self.relu), ptr(dy), None, 0, m, kp, st), 'tok_bn_bwd_apply')
        out.grad = None
        ctot = sum(t.cp for t in srcs)
        wd = self.pk.dgrad.view(ctot, kp) if self.pk.dgrad is not None else None
        # d(y_j) = up_j^T d(y): the transposes of the interpolations over ALL kp channels of d(y) — one pass for the three
        # low-resolution sources (source 0: identity)
        dys = [dy] + [torch.empty((n, d.h, d.w, kp), dtype=BF16, device=dy.device) for d in self.descs[1:]]
        low = []
        for j in range(1, 4):
            low += [ptr(dys[j]), self.descs[j].h, self.descs[j].w] if j < len(dys) else [None, 1, 1]
        _C.check(lib.tok_bilinear_bwd_multi(ptr(dy), n, h, w, kp, *low, st), 'tok_bilinear_bwd_multi')
        for j, (t, d) in enumerate(zip(srcs, self.descs)):
            if t.requires_grad:
                tgt, acc = grad_target(t)
                off = self.offs[j]
                _C.check(lib.tok_conv_dgrad(d, ptr(dys[j]), ptr(wd[off:off + t.cp]), ptr(tgt), acc, st), 'tok_conv_dgrad')
        if not w_need:
            return
        k_real = conv.weight.shape[0]

        def run_wgrads():
            st2 = stream_ptr()
            slot, mode = param_grad_target(conv.weight)
            slot2 = torch.as_strided(slot, (k_real, ctot), (ctot, 1), slot.storage_offset())
            keep = []
            for j, (t, d) in enumerate(zip(srcs, self.descs)):
                ws_bytes = lib.tok_conv_wgrad_ws_bytes(d)
                ws = torch.empty(max(ws_bytes // 4, 1), dtype=F32, device=dy.device)
                dwj = torch.empty((k_real, t.cp), dtype=F32, device=dy.device)
                _C.check(lib.tok_conv_wgrad(d, ptr(t.data), ptr(dys[j]), ptr(dwj), k_real, t.cp, ptr(ws), ws_bytes, 0, st2),
                         'tok_conv_wgrad')
                off = self.offs[j]
                if mode == 1:
                    slot2[:, off:off + t.cp].add_(dwj)
                else:
                    slot2[:, off:off + t.cp].copy_(dwj)
                keep += [ws, dwj]
            commit_param_grad(conv.weight, slot, mode)
            return keep

        side = (NECK_WGRAD_SIDE and EF.WGRAD_SIDE_STREAM and EF._side_for_tag(self.stream_tag, self.region) and dy.is_cuda
                and self.region is not None and not torch.cuda.is_current_stream_capturing())
        if side:
            with self.region.fork_side([t.data for t in srcs] + dys):
                self.region.keep_until_join(*run_wgrads())
        else:
            run_wgrads()


def commuted_ok(srcs: List[TTensor], size: Tuple[int, int], conv: nn.Module, bn: Optional[nn.BatchNorm2d]) -> bool:
    if not NECK_COMMUTE or not isinstance(conv, nn.Conv2d) or bn is None or len(srcs) < 2 or len(srcs) > 4:
        return False
    if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.padding != (0, 0) or conv.bias is not None \
            or conv.groups != 1 or conv.dilation != (1, 1):
        return False
    if not (bn.training or bn.running_mean is None) or bn.momentum is None:
        return False
    x0 = srcs[0]
    if x0.data.dim() != 4 or (x0.shape[1], x0.shape[2]) != (int(size[0]), int(size[1])):
        return False
    if any(t.c != t.cp or t.shape[0] != x0.shape[0] for t in srcs):
        return False
    k = conv.out_channels
    return k == pad8(k) and sum(t.c for t in srcs) == conv.in_channels and conv.weight.permute(0, 2, 3, 1).is_contiguous()


def upsample_concat_conv_bn_relu(region: Region, srcs: List[TTensor], size: Tuple[int, int], conv: nn.Module,
                                 bn: nn.BatchNorm2d, relu: bool = True) -> TTensor:
    """relu(bn(conv1x1(cat([interpolate(s, size, 'bilinear') for s in srcs], 1)))) — in the commuted order where it is served
    (pointwise filter without bias, BatchNorm on batch statistics, channel widths that are multiples of 8, first source at the
    target size), in the direct order otherwise."""
    if not commuted_ok(srcs, size, conv, bn):
        feats = ER.bilinear_concat(region, srcs, size)
        return EF.conv_bn_act(region, feats, conv, bn, relu=relu)
    await_ready(*srcs)
    lib, st = _C.lib(), stream_ptr()
    kp = conv.out_channels
    ctot = conv.in_channels
    x_need = region.grad_mode and any(t.requires_grad for t in srcs)
    training = region.grad_mode and (conv.weight.requires_grad or x_need or bn.weight.requires_grad)
    pk = EF.get_packs(conv.weight, None, kp, 1, ctot, want_dgrad=x_need, refresh=True)
    wf = pk.fwd.view(kp, ctot)
    dev = srcs[0].data.device
    n, h, w, _ = srcs[0].shape
    m = n * h * w
    ys, descs, offs = [], [], []
    off = 0
    for t in srcs:
        d = EF._conv_desc(t, kp, 1, 1, 1, 0)
        yj = torch.empty((d.n, d.p, d.q, kp), dtype=BF16, device=dev)
        wj = wf[:, off:off + t.cp].contiguous()          # the filter columns of this source: [kp][c_j]
        _C.check(lib.tok_conv_fwd(d, ptr(t.data), ptr(wj), None, ptr(yj), None, st), 'tok_conv_fwd')
        ys.append(yj)
        descs.append(d)
        offs.append(off)
        off += t.cp
    rows = lib.tok_bilinear_sum_stats_rows(n, h, w, kp)
    stats = torch.empty((2, rows, kp), dtype=F32, device=dev)
    low = []
    for j in range(1, 4):
        low += [ptr(ys[j]), descs[j].h, descs[j].w] if j < len(ys) else [None, 1, 1]
    y = ys[0]
    _C.check(lib.tok_bilinear_sum_stats(ptr(y), *low, n, h, w, kp, ptr(y), ptr(stats), st), 'tok_bilinear_sum_stats')
    vec = torch.empty((4, kp), dtype=F32, device=dev)
    scale, shift, mean, rstd = vec[0], vec[1], vec[2], vec[3]
    track = bn.training and bn.track_running_stats and bn.running_mean is not None
    _C.check(lib.tok_bn_finalize(ptr(stats), rows, m, kp, bn.num_features, ptr(bn.weight), ptr(bn.bias),
                                 ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                 ptr(bn.num_batches_tracked) if track else None, float(bn.momentum), float(bn.eps),
                                 ptr(mean), ptr(rstd), ptr(scale), ptr(shift), st), 'tok_bn_finalize')
    out_data = torch.empty_like(y)
    mask = torch.empty((m, kp // 8), dtype=torch.uint8, device=dev) if (relu and training) else None
    _C.check(lib.tok_bn_act_fwd(ptr(y), ptr(scale), ptr(shift), None, int(relu), ptr(out_data), ptr(mask), m, kp, st),
             'tok_bn_act_fwd')
    out = TTensor(out_data, kp, requires_grad=bool(training))
    if training:
        node = _CommutedNeckNode()
        node.x, node.out, node.shortcut, node.y = None, out, None, y
        node.conv, node.bn, node.desc, node.pk = conv, bn, descs[0], pk
        node.relu, node.batch_stats, node.mask = relu, True, mask
        node.mean, node.rstd, node.scale, node.shift = mean, rstd, scale, shift
        node.srcs, node.descs, node.offs = list(srcs), descs, offs
        node.sub_capable = False
        out.node = node
        for t in srcs:
            if t.requires_grad:
                t.uses += 1
        region.add(node)
    return out
