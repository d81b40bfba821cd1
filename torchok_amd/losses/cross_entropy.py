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


class UpsampledLogits(torch.Tensor):
    """What SegmentationHead returns in training: the (N, C, H, W) result of `F.interpolate(low, size, mode='bilinear')`
    (reference heads/segmentation/base.py:37) WITHOUT having computed it.  CrossEntropyLoss recognises the object and runs the
    fused kernels on the low-resolution logits (tok_upsample_ce_fwd / _bwd: the full-resolution tensor and its gradient —
    2 x 604 MB per HRNet-W48 step at 512x1024, batch 24 — are never written).  Anything else that touches it (a metric, a
    Dice loss, user code) gets the real tensor: every torch function applied to it first materialises the interpolation
    through the engine's bilinear unit (with its autograd edge to the low-resolution logits) — metadata queries (shape, dtype,
    device, dim, size, requires_grad) excepted."""

    @staticmethod
    def __new__(cls, low: Tensor, size):
        n, c = low.shape[:2]
        r = torch.Tensor._make_wrapper_subclass(cls, (n, c, int(size[0]), int(size[1])), dtype=low.dtype, device=low.device,
                                                requires_grad=False)
        r._low, r._size, r._full = low, (int(size[0]), int(size[1])), None
        return r

    def __init__(self, low: Tensor, size):
        pass

    def materialize(self) -> Tensor:
        if self._full is None:
            from .. import engine
            from ..engine import resample as ER
            with engine.region() as r:
                self._full = r.output(ER.bilinear_resize(r, r.input(self._low), self._size))
        return self._full

    # what the storage-less wrapper may answer itself: shape / dtype / device queries only.  Anything that exposes memory
    # (data_ptr, stride, is_contiguous, .data, .grad ...) materialises first: the wrapper has no storage, its strides describe
    # a contiguous NCHW tensor while the real one is channels-last with a padded pitch.
    _META = {'dim', 'size', 'numel', 'is_floating_point', '__len__', 'ndimension', 'nelement', 'element_size', 'is_complex',
             'get_device', '__repr__', '__str__', '__format__'}
    # (the autograd properties — requires_grad, grad_fn, is_leaf, output_nr, _version — are NOT answered here: the wrapper is a
    #  leaf that requires no gradient, the materialised tensor hangs on the head's graph; a user loss that tests
    #  `input.requires_grad` or hands the input to its own autograd.Function must see the real tensor's answers)
    _META_PROPS = {'shape', 'dtype', 'device', 'ndim', 'is_cuda', 'is_cpu', 'layout', 'names',
                   'is_sparse', 'is_quantized', 'is_meta', 'is_mkldnn', 'is_nested', 'is_sparse_csr', 'is_xpu',
                   'is_mps', 'is_xla', 'is_vulkan', 'is_ipu', 'is_maia', 'is_mtia', 'name'}

    def __repr__(self, *, tensor_contents=None):
        return f'UpsampledLogits(low={tuple(self._low.shape)}, size={self._size}, materialized={self._full is not None})'

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, '__name__', '')
        if name in cls._META or (name == '__get__' and
                                 getattr(getattr(func, '__self__', None), '__name__', None) in cls._META_PROPS):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def real(a):
            if isinstance(a, UpsampledLogits):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*[real(a) for a in args], **{k: real(v) for k, v in kwargs.items()})


    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by code that bypasses __torch_function__ (C++ callers): same answer, the real tensor
        def real(a):
            if isinstance(a, UpsampledLogits):
                return a.materialize()
            if isinstance(a, (list, tuple)):
                return type(a)(real(v) for v in a)
            return a
        return func(*[real(a) for a in args], **{k: real(v) for k, v in (kwargs or {}).items()})


def materialized(x):
    """The real tensor behind an UpsampledLogits (anything else is returned as is).  `torch.autograd.Function.apply` does not
    go through `__torch_function__`: it would see the wrapper (no grad_fn, requires_grad False) and return a loss that is cut
    off from the head — every loss that calls a Function on its `input` passes it through here first."""
    return x.materialize() if isinstance(x, UpsampledLogits) else x


class _UpsampleCE(torch.autograd.Function):
    """CrossEntropyLoss(F.interpolate(low, size, 'bilinear'), target) from the low-resolution logits (csrc/loss.hip)."""

    @staticmethod
    def forward(ctx, low: Tensor, target: Tensor, size, ignore_index: int):
        require_device(low)
        n, classes, hs, ws = low.shape
        hd, wd = size
        zp = low.detach().permute(0, 2, 3, 1)
        ld = zp.stride(2)
        if not (zp.dtype == BF16 and zp.stride(3) == 1 and ld >= classes and ld % 8 == 0 and zp.stride(1) == ws * ld
                and zp.stride(0) == hs * ws * ld and zp.data_ptr() % 16 == 0):
            ld = pad8(classes)
            buf = torch.zeros((n, hs, ws, ld), dtype=BF16, device=low.device)
            buf[..., :classes] = zp
            zp = buf[..., :classes]
        z = torch.as_strided(zp, (n, hs, ws, ld), (hs * ws * ld, ws * ld, ld, 1))
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.to(torch.int64).contiguous()
        dev = z.device
        rows = n * hd * wd
        lse = torch.empty(rows, dtype=torch.float32, device=dev)
        row_loss = torch.empty(rows, dtype=torch.float32, device=dev)
        loss = torch.empty(_C.TOK_CE_LOSS_FLOATS, dtype=torch.float32, device=dev)
        _C.check(_C.lib().tok_upsample_ce_fwd(ptr(z), n, hs, ws, classes, ld, hd, wd, ptr(target), ignore_index, ptr(lse),
                                              ptr(row_loss), ptr(loss), stream_ptr()), 'tok_upsample_ce_fwd')
        ctx.z, ctx.target, ctx.lse, ctx.loss, ctx.ignore_index = z, target, lse, loss, ignore_index
        ctx.geom = (n, hs, ws, classes, ld, hd, wd)
        ctx.in_dtype = low.dtype
        n_valid = loss[1]
        ctx.mark_non_differentiable(n_valid)
        return loss[0], n_valid

    @staticmethod
    def backward(ctx, g, _unused=None):
        n, hs, ws, classes, ld, hd, wd = ctx.geom
        gs = g.detach().to(torch.float32).reshape(1).contiguous()
        d = torch.empty((n, hs, ws, ld), dtype=BF16, device=ctx.z.device)
        _C.check(_C.lib().tok_upsample_ce_bwd(ptr(ctx.z), n, hs, ws, classes, ld, hd, wd, ptr(ctx.target), ctx.ignore_index,
                                              ptr(ctx.lse), ptr(ctx.loss), ptr(gs), ptr(d), 0, stream_ptr()),
                 'tok_upsample_ce_bwd')
        ctx.z = ctx.target = ctx.lse = ctx.loss = None
        if ld == pad8(classes) != classes:
            mark_padded(d)
        d = d[..., :classes].permute(0, 3, 1, 2)
        if ctx.in_dtype != BF16:
            d = d.to(ctx.in_dtype)
        return d, None, None, None


# TOK_FUSE_UPSAMPLE_CE=0: SegmentationHead returns the interpolated tensor itself (the round-3 path)
FUSE_UPSAMPLE_CE = os.environ.get('TOK_FUSE_UPSAMPLE_CE', '1') == '1'


@LOSSES.register_class
class CrossEntropyLoss(nn.Module):
    consumes_lazy_logits = True     # JointLoss hands the UpsampledLogits through instead of materialising it

    def __init__(self, weight=None, size_average=None, ignore_index: int = -100, reduce=None,
                 reduction: str = 'mean', label_smoothing: float = 0.0):
        super().__init__()
        if weight is not None or reduction not in ('mean', 'sum') or size_average is not None or reduce is not None:
            raise NotImplementedError("torchok_amd CrossEntropyLoss: reduction 'mean' or 'sum', no class weights")
        if not 0.0 <= label_smoothing <= 1.0:
            raise ValueError(f'label_smoothing must be between 0.0 and 1.0. Got: {label_smoothing}')
        self.ignore_index, self.label_smoothing, self.reduction = ignore_index, label_smoothing, reduction

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if isinstance(input, UpsampledLogits):
            low, size = input._low, input._size
            ld = low.permute(0, 2, 3, 1).stride(2) if low.dim() == 4 else 0
            if (self.label_smoothing == 0.0 and input._full is None and target.shape == (low.shape[0],) + size
                    and _C.lib().tok_upsample_ce_serves(low.shape[1], max(ld, pad8(low.shape[1])))):
                loss, n_valid = _UpsampleCE.apply(low, target, size, self.ignore_index)
                return loss if self.reduction == 'mean' else loss * n_valid
            input = input.materialize()
        if input.dim() not in (2, 4):
            raise NotImplementedError('torchok_amd CrossEntropyLoss: (N, C) or (N, C, H, W) logits')
        if input.dim() == 4 and target.shape != (input.shape[0],) + tuple(input.shape[2:]):
            raise ValueError(f'Expected target of shape (N, H, W) for (N, C, H, W) logits, got {tuple(target.shape)}')
        loss, n_valid = _SoftmaxCE.apply(input, target, self.ignore_index, self.label_smoothing)
        # 'sum' = mean over the non-ignored rows * their number
        return loss if self.reduction == 'mean' else loss * n_valid
