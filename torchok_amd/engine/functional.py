"""Fused units recorded on the engine tape.  Each function launches the forward kernels through
the C ABI and (in grad mode) appends a Node whose ``backward`` launches the backward kernels.

Unit                      replaces (reference / [timm] / ATen)
------------------------  -------------------------------------------------------------------
conv_bn_act               conv2d -> batch_norm -> (+ shortcut) -> relu   ([timm] Bottleneck /
                          BasicBlock / downsample_conv; resnet.py:488-490; convbnact.py:48-53)
max_pool_3x3_s2           nn.MaxPool2d(3, 2, 1)                          (resnet.py:510)
global_avg_pool           SelectAdaptivePool2d('avg', flatten=True)      (pooling.py:7-12)
linear                    nn.Linear                                      (linear_head.py:31)
"""
import weakref
from typing import Optional

import os

import torch
from torch import nn

from .. import _C
from .core import _ms as _core_ms
from .core import (BF16, Node, Region, TTensor, await_ready, commit_param_grad, donate_grad, grad_target,
                   is_last_contribution, pad8, param_grad_target, ptr, stream_ptr)

F32 = torch.float32


def core_hooks():
    from . import core
    return core.param_grad_hooks


# ---- packed bf16 operands of the fp32 master weights ------------------------------------------------

class _Packs:
    """bf16 MFMA operands derived from one fp32 master weight: the forward pack [Kp][R][Sp][Cp]
    and the dgrad pack [Cp][R][S][Kp] (flipped taps).  Re-derived whenever `refresh` is asked
    (every training forward: the optimizer has moved the master since)."""
    __slots__ = ('fwd', 'dgrad', 'bias', 'key', 'synced', 'dims')

    def __init__(self):
        self.fwd = None
        self.dgrad = None
        self.bias = None
        self.key = None
        self.synced = None   # weight._version the packs were derived at (None: unknown / stale)
        self.dims = None     # (k, r, s, c, kp, sp, cp)


_packs = {}  # id(weight) -> (weakref(weight), _Packs); Tensor.__eq__ rules out a WeakKeyDictionary


def _packs_for(weight) -> _Packs:
    ent = _packs.get(id(weight))
    if ent is not None and ent[0]() is weight:
        return ent[1]
    pk = _Packs()
    key = id(weight)
    _packs[key] = (weakref.ref(weight, lambda _r, key=key: _packs.pop(key, None)), pk)
    return pk


def _krsc(weight: torch.Tensor):
    """(K, R, S, C) of a conv (K,C,R,S) or linear (K,C) master and a guarantee that its
    physical layout is [k][r][s][c]."""
    if weight.dim() == 2:
        if not weight.is_contiguous():
            raise RuntimeError('linear weight must be contiguous')
        return weight.shape[0], 1, 1, weight.shape[1]
    k, c, r, s = weight.shape
    if not weight.permute(0, 2, 3, 1).is_contiguous():
        # re-home the master in channels-last order once (logical shape / state_dict unchanged)
        weight.data = weight.data.contiguous(memory_format=torch.channels_last)
        if not weight.permute(0, 2, 3, 1).is_contiguous():  # degenerate dims: force strides
            weight.data = torch.as_strided(weight.data.permute(0, 2, 3, 1).contiguous().view(-1),
                                           (k, c, r, s), (r * s * c, 1, s * c, c))
    return k, r, s, c


def get_packs(weight: nn.Parameter, bias: Optional[nn.Parameter], kp: int, sp: int, cp: int,
              want_dgrad: bool, refresh: bool) -> _Packs:
    pk = _packs_for(weight)
    k, r, s, c = _krsc(weight)
    key = (weight.data_ptr(), kp, sp, cp, weight.device)
    lib = _C.lib()
    st = stream_ptr()
    global _pack_generation
    # packs stay valid while the master is untouched: torch-visible writes bump `_version`; the fused
    # optimizers (whose kernels do not) refresh every pack themselves right after the step (repack_after_step)
    stale = pk.key != key or (refresh and pk.synced != weight._version)
    if want_dgrad and (pk.fwd is None or pk.dgrad is None or stale):
        if pk.fwd is None or pk.key != key:
            pk.fwd = torch.empty((kp, r, sp, cp), dtype=BF16, device=weight.device)
            _pack_generation += 1
        if pk.dgrad is None or pk.key != key:
            pk.dgrad = torch.empty((cp, r, s, kp), dtype=BF16, device=weight.device)
            _pack_generation += 1
        _C.check(lib.tok_pack_weight_both(ptr(weight), k, r, s, c, ptr(pk.fwd), kp, sp, cp, ptr(pk.dgrad), st),
                 'tok_pack_weight_both')
        pk.synced = weight._version
    elif pk.fwd is None or stale:
        if pk.fwd is None or pk.key != key:
            pk.fwd = torch.empty((kp, r, sp, cp), dtype=BF16, device=weight.device)
            _pack_generation += 1
        if pk.key != key:
            pk.dgrad = None
        _C.check(lib.tok_pack_weight_fwd(ptr(weight), k, r, s, c, ptr(pk.fwd), kp, sp, cp, st),
                 'tok_pack_weight_fwd')
        if pk.dgrad is not None:       # an older dgrad pack is no longer in step with the master
            pk.dgrad = None
            _pack_generation += 1
        pk.synced = weight._version
    pk.dims = (k, r, s, c, kp, sp, cp)
    if bias is not None:
        if kp == k:
            pk.bias = bias.detach()
        else:
            if pk.bias is None or pk.bias.shape[0] != kp:
                pk.bias = torch.zeros(kp, dtype=F32, device=weight.device)
            pk.bias[:k] = bias.detach()
    else:
        pk.bias = None
    pk.key = key
    return pk


WGRAD_SIDE_STREAM = os.environ.get('TOK_WGRAD_SIDE', '1') == '1'
# which weight gradients go to the side stream: 'all', or only the LDS/MFMA-bound ones ('3x3': filters larger than 1x1),
# whose resource profile complements the HBM-bound main chain
WGRAD_AFTER_DGRAD = os.environ.get('TOK_WGRAD_AFTER_DGRAD', '0') == '1'   # measured: 22.0 vs 21.1 ms/step — the later start costs more
WGRAD_SIDE_MAX_ROWS = int(os.environ.get('TOK_WGRAD_SIDE_MAX_ROWS', '100000'))
WGRAD_SIDE_WHICH = os.environ.get('TOK_WGRAD_SIDE_WHICH', '3x3')   # measured (ResNet-50, unit-3 fusion on): all 21.58, 3x3 21.45 ms/step
# long-M pointwise weight gradients stay on the main stream when they are HBM-bound like the chain they would fight (ResNet-50: 51-102
# MACs per operand element), and go to the side stream when they are MFMA-bound (HRNet-W48's 720 -> 720 neck convolution over 786 432
# pixels: 360): HRNet-W48 70.65 -> 69.83 ms/step, ResNet-50 / SwinV2-T unchanged (profiles/r05_side_stream_resweep.txt)
WGRAD_SIDE_MIN_INTENSITY = float(os.environ.get('TOK_WGRAD_SIDE_MIN_INTENSITY', '128'))


def _pointwise_wgrad_is_mfma_bound(k, c) -> bool:
    return (k * c) / float(k + c) >= WGRAD_SIDE_MIN_INTENSITY


# Only units of the MAIN stream fork their weight gradients to the side stream (round 6).  A unit recorded on a branch stream
# (HRNet's low-resolution branches, SwinV2's position-bias chain) keeps them on its own stream: with three branch streams + the
# side stream there are more streams than hardware queues (two share one), and the side queue — every weight gradient of every
# branch in one FIFO — was 33 ms busy beside a 46-ms HRNet-W48 backward.  HRNet-W48 B=24: 67.1 -> 66.1 ms/step (nobody forks:
# 66.1 as well; only the branches fork: 67.9).  TOK_WGRAD_SIDE_TAGS=<comma-separated stream tags that fork> overrides ("0,1,2,3":
# rounds 1-5).
_SIDE_TAGS = {int(v) for v in os.environ.get('TOK_WGRAD_SIDE_TAGS', '0').split(',') if v != ''}
_SIDE_TAGS_FORCED = 'TOK_WGRAD_SIDE_TAGS' in os.environ


def _side_for_tag(tag, region=None) -> bool:
    # ... and nobody forks in a region whose main stream and branch streams alone fill the four hardware queues (HRNet-W48: main +
    # three branches; a fifth stream shares a queue with one of them: 65.4 vs 65.9 ms/step without the side stream, same box)
    if int(tag or 0) not in _SIDE_TAGS:
        return False
    return _SIDE_TAGS_FORCED or region is None or len(getattr(region, '_streams', ())) < 3


FUSE_BN_FINALIZE = os.environ.get('TOK_FUSE_BN_FINALIZE', '0') == '1'    # tok_conv_*_bn ("last workgroup finalizes") measured slower than the stand-alone finalize launches, see DESIGN.md §4
_ticket_rings = {}


def _ticket_counters(device):
    """Pointer to 64 zeroed device ints for one fused-finalize launch.  A ring of 256 slots per device: kernels
    leave their counters zero, and no 256 such launches are ever in flight at once."""
    ring = _ticket_rings.get(device)
    if ring is None:
        ring = _ticket_rings[device] = [torch.zeros(256 * 64, dtype=torch.int32, device=device), 0]
    ring[1] = (ring[1] + 1) & 255
    return ring[0].data_ptr() + ring[1] * 256


_pack_generation = 0   # bumped whenever a pack buffer is (re)allocated: invalidates cached batch tables


def invalidate_packs():
    """Force every bf16 operand pack to be re-derived at its next use.  Needed only after writing a weight
    behind torch's back (`p.data.mul_(...)`, a custom kernel): such writes do not bump `p._version`."""
    for _, pk in list(_packs.values()):
        pk.synced = None


def repack_after_step(params, cache: dict, arena_sig) -> None:
    """Refresh the packs of every weight in `params` with ONE launch (tok_pack_weights_batched).  Called by
    the fused optimizers at the end of step(): their kernels update the masters without touching
    `_version`, so this is what keeps packs and masters in step."""
    import ctypes
    sig = (_pack_generation, arena_sig)
    if cache.get('sig') != sig:
        items, entries, skipped = [], [], []
        block = 0
        for p in params:
            ent = _packs.get(id(p))
            if ent is None or ent[0]() is not p:
                continue
            pk = ent[1]
            if pk.fwd is None or pk.key is None or pk.dims is None or pk.key[0] != p.data_ptr():
                skipped.append(pk)
                continue
            k, r, s, c, kp, sp, cp = pk.dims
            it = _C.PackItem(p.data_ptr(), pk.fwd.data_ptr(), pk.dgrad.data_ptr() if pk.dgrad is not None else None,
                             k, r, s, c, kp, sp, cp, block)
            nb = _C.lib().tok_pack_item_blocks(ctypes.byref(it))
            if nb <= 0:                       # a pack of 2^31 elements or more: refreshed on its own by the forward
                skipped.append(pk)
                continue
            block += nb
            items.append(it)
            entries.append((p, pk))
        dev = None
        if items:
            arr = (_C.PackItem * len(items))(*items)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = host.to(entries[0][0].device)
        cache.clear()
        cache.update(sig=sig, dev=dev, n=len(items), blocks=block, entries=entries, skipped=skipped)
    for pk in cache['skipped']:
        pk.synced = None
    if cache['n']:
        _C.check(_C.lib().tok_pack_weights_batched(ptr(cache['dev']), cache['n'], cache['blocks'], stream_ptr()),
                 'tok_pack_weights_batched')
        for p, pk in cache['entries']:
            pk.synced = p._version


def _conv_desc(x: TTensor, k_pad: int, r: int, s: int, stride: int, pad: int) -> _C.ConvDesc:
    if x.data.dim() == 2:      # (N, Cp) rows: the 1x1 / h = w = 1 case (Linear)
        n, cp = x.shape
        h = w = 1
    else:
        n, h, w, cp = x.shape
    p = (h + 2 * pad - r) // stride + 1
    q = (w + 2 * pad - s) // stride + 1
    return _C.ConvDesc(n, h, w, cp, k_pad, r, s, p, q, stride, pad, 8 if cp == 4 else s)


def _check_conv(conv: nn.Conv2d):
    if conv.groups != 1 or tuple(conv.dilation) != (1, 1):
        raise NotImplementedError('torchok_amd conv: groups == 1 and dilation == 1 only')
    if conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1] \
            or conv.padding[0] != conv.padding[1] or isinstance(conv.padding, str):
        raise NotImplementedError('torchok_amd conv: square kernel / stride / padding only')
    if conv.padding_mode != 'zeros':
        raise NotImplementedError('torchok_amd conv: zero padding only')


# ---- conv + bn + (add) + relu ------------------------------------------------------------------------

class _ConvBnActNode(Node):
    needs_backward = True

    def __init__(self):
        self.x = self.out = self.shortcut = None
        self.y = self.mask = None
        self.region = None
        self.fused_partial = None   # (partial, rows) when a consumer's dgrad epilogue did our BN-bwd reduce
        self.fused_coef = None      # apply coefficients when that dgrad also finalized (tok_conv_dgrad_bn)
        self.coef = None
        self.pool = None            # tap indices of the fused 3x3/s2 max-pool: `out` is the POOLED map, z was never stored
        self.ypool = None           # raw conv output at the winning taps (pooled-domain BatchNorm-backward sums)

    def release(self):
        self.x = self.out = self.shortcut = None
        self.y = self.pk = self.mask = self.fused_partial = self.fused_coef = self.coef = self.pool = self.ypool = None
        self.mean = self.rstd = self.scale = self.shift = None

    def wants_fused_bwd_stats(self) -> bool:
        """Can the kernel that completes d(out) also reduce sum(dz), sum(dz*y) for this unit?"""
        return self.bn is not None and self.batch_stats and (not self.relu or self.mask is not None) and self.pool is None

    def _finalize_bwd(self, lib, st, g, mask, m, kp, g_need, b_need):
        """sum(dz), sum(dz*xhat) -> dgamma, dbeta, apply coefficients (stand-alone reduce / finalize launches)."""
        bn = self.bn
        dzy = 0
        if self.fused_partial is not None:
            partial, rows = self.fused_partial   # reduced by the dgrad that completed d(out)
            dzy = 1
        else:
            rows = lib.tok_bn_bwd_rows(m, kp)
            if self.pool is not None and self.ypool is not None:
                # sums over the pooled elements (each feeds exactly one position): 3 pooled-size reads, no gather
                mp = self.out.data.numel() // kp
                rows = lib.tok_bn_bwd_rows(mp, kp)
                partial = torch.empty((2, rows, kp), dtype=F32, device=g.device)
                _C.check(lib.tok_bn_pool_bwd_reduce_pooled(ptr(g), ptr(self.out.data), ptr(self.ypool), ptr(self.mean),
                                                           ptr(self.rstd), mp, kp, ptr(partial), st),
                         'tok_bn_pool_bwd_reduce_pooled')
            elif self.pool is not None:
                partial = torch.empty((2, rows, kp), dtype=F32, device=g.device)
                n_, h_, w_, _ = self.y.shape
                _C.check(lib.tok_bn_pool_bwd_reduce(ptr(g), ptr(self.pool), ptr(self.y), ptr(self.scale), ptr(self.shift),
                                                    ptr(self.mean), ptr(self.rstd), n_, h_, w_, kp, ptr(partial), st),
                         'tok_bn_pool_bwd_reduce')
            else:
                partial = torch.empty((2, rows, kp), dtype=F32, device=g.device)
                _C.check(lib.tok_bn_bwd_reduce(ptr(g), ptr(self.y), ptr(mask), ptr(self.scale), ptr(self.shift),
                                               ptr(self.mean), ptr(self.rstd), int(self.relu), m, kp, ptr(partial), st),
                         'tok_bn_bwd_reduce')
        coef = torch.empty((3, kp), dtype=F32, device=g.device)
        gs, gm = param_grad_target(bn.weight) if g_need else (None, 0)
        bs, bm = param_grad_target(bn.bias) if b_need else (None, 0)
        if gm == 2 or bm == 2 or (gm != bm and g_need and b_need):
            # rare mixed state: run the accumulate-free form and fix up on the host side
            gacc = torch.empty_like(gs) if g_need else None
            bacc = torch.empty_like(bs) if b_need else None
            _C.check(lib.tok_bn_bwd_finalize(ptr(partial), rows, m, kp, bn.num_features, ptr(bn.weight), ptr(self.mean),
                                             ptr(self.rstd), ptr(gacc), ptr(bacc), ptr(coef), 0, dzy, st),
                     'tok_bn_bwd_finalize')
            for p_, acc_ in ((bn.weight, gacc), (bn.bias, bacc)):
                if acc_ is not None:
                    if p_.grad is None:
                        p_.grad = acc_
                    else:
                        p_.grad.add_(acc_)
                    for h_ in core_hooks():
                        h_(p_)
        else:
            def commit():
                if g_need:
                    commit_param_grad(bn.weight, gs, gm)
                if b_need:
                    commit_param_grad(bn.bias, bs, bm)
            _C.check(lib.tok_bn_bwd_finalize(ptr(partial), rows, m, kp, bn.num_features, ptr(bn.weight), ptr(self.mean),
                                             ptr(self.rstd), ptr(gs), ptr(bs), ptr(coef),
                                             1 if (gm == 1 or bm == 1) else 0, dzy, st), 'tok_bn_bwd_finalize')
            commit()
        self.coef = coef
        return None

    def _wgrad_goes_side(self, g, m) -> bool:
        conv = self.conv
        if conv.weight.dim() != 4:
            r = s = 1
        else:
            r, s = conv.weight.shape[2], conv.weight.shape[3]
        side_ok = (WGRAD_SIDE_WHICH == 'all' or (r * s > 1 and WGRAD_SIDE_WHICH != '1x1') or
                   (m < WGRAD_SIDE_MAX_ROWS and (r * s == 1 or WGRAD_SIDE_WHICH != '1x1')) or
                   (r * s == 1 and _pointwise_wgrad_is_mfma_bound(conv.weight.shape[0], conv.weight.shape[1])))   # == launch_wgrad's side_ok
        return bool(WGRAD_SIDE_STREAM and side_ok and _side_for_tag(self.stream_tag, self.region) and g.is_cuda and self.region is not None and not WGRAD_AFTER_DGRAD
                    and not torch.cuda.is_current_stream_capturing())

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        out: TTensor = self.out
        g = out.grad
        if g is None:
            return
        apply_event = None
        conv, bn, d = self.conv, self.bn, self.desc
        m, kp = self.y.numel() // self.y.shape[-1], self.y.shape[-1]
        x: TTensor = self.x
        w_need = conv.weight.requires_grad
        x_need = x.requires_grad
        bias_need = conv.bias is not None and conv.bias.requires_grad
        sc: Optional[TTensor] = self.shortcut

        if bn is not None:
            g_need = bn.weight.requires_grad
            b_need = bn.bias.requires_grad
            sc_need = sc is not None and sc.requires_grad
            mask = self.mask if self.relu else None
            need_dy = w_need or x_need or bias_need
            if self.batch_stats:
                if self.fused_coef is not None:
                    # the dgrad that completed d(out) reduced AND finalized (tok_conv_dgrad_bn): dgamma / dbeta are
                    # already in their slots, only the apply coefficients are needed here
                    coef = self.fused_coef
                else:
                    self._finalize_bwd(lib, st, g, mask, m, kp, g_need, b_need)
                    coef = self.coef
            else:
                # eval-mode BN: y -> out is a fixed affine map: dy = scale * dz, dgamma/dbeta unsupported
                if g_need or b_need:
                    raise NotImplementedError('gradients of BatchNorm affine parameters in eval mode')
                coef = torch.zeros((3, kp), dtype=F32, device=g.device)
                coef[0] = self.scale
            need_dy = w_need or x_need or bias_need
            if need_dy or sc_need:
                dy = torch.empty_like(self.y)
                if LAUNCH_EVENTS and w_need and self.pool is None and self._wgrad_goes_side(g, m):
                    # the weight gradient will be forked to the side stream behind THIS apply pass: the pass carries the
                    # completion event itself (no event-record packet on the main queue)
                    apply_event = self.region.raw_event()     # armed right in front of the launch that carries it (below)
                ds_ptr, ds_acc = None, 0
                if sc_need:
                    if sc.grad is None and out.grad_owned:
                        # in-place mask of the incoming gradient, then donate it to the shortcut
                        ds_ptr = ptr(g)
                        donate_grad(sc, g)
                    else:
                        tgt, ds_acc = grad_target(sc)
                        ds_ptr = ptr(tgt)
                if self.pool is not None:
                    n_, h_, w_, _ = self.y.shape
                    _C.check(lib.tok_bn_pool_bwd_apply(ptr(g), ptr(self.pool), ptr(self.y), ptr(self.scale), ptr(self.shift),
                                                       ptr(coef), n_, h_, w_, kp, ptr(dy), st), 'tok_bn_pool_bwd_apply')
                else:
                    if apply_event is not None:
                        lib.tok_next_launch_event(apply_event)
                    try:
                        _C.check(lib.tok_bn_bwd_apply(ptr(g), ptr(self.y), ptr(mask), ptr(self.scale),
                                                      ptr(self.shift), ptr(coef), int(self.relu), ptr(dy), ds_ptr,
                                                      ds_acc, m, kp, st), 'tok_bn_bwd_apply')
                    finally:
                        if apply_event is not None:
                            lib.tok_next_launch_event(None)   # never left armed for an unrelated later launch
            else:
                dy = None
        else:
            dy = g  # plain conv (+bias): the output gradient IS dy
        out.grad = None
        if dy is None:
            return
        # the bias gradient (column sums of dy) rides the weight-gradient kernel where that serves the layer
        bias_in_wgrad = bool(bias_need and w_need and BIAS_IN_WGRAD and lib.tok_conv_wgrad_bias_ok(d))
        if bias_need and not bias_in_wgrad:
            bs, bm = param_grad_target(conv.bias)
            if m > 4096 and kp == conv.bias.shape[0]:
                # tall dy (a conv / token GEMM bias): coalesced row-chunk partials, then a fixed-order fold
                nrows = lib.tok_colsum_partial_rows(m, kp)
                part = torch.empty((nrows, kp), dtype=F32, device=g.device)
                _C.check(lib.tok_colsum_partial(ptr(dy), m, kp, ptr(part), st), 'tok_colsum_partial')
                _C.check(lib.tok_colsum_f32(ptr(part), nrows, kp, ptr(bs), 1 if bm == 1 else 0, st), 'tok_colsum_f32')
            else:
                _C.check(lib.tok_colsum(ptr(dy), m, kp, conv.bias.shape[0], ptr(bs), 1 if bm == 1 else 0, st),
                         'tok_colsum')
            commit_param_grad(conv.bias, bs, bm)
        def launch_wgrad():
            k, r, s, c = _krsc(conv.weight)
            ws_bytes = lib.tok_conv_wgrad_bias_ws_bytes(d) if bias_in_wgrad else lib.tok_conv_wgrad_ws_bytes(d)

            def run_wgrad():
                ws = torch.empty(max(ws_bytes // 4, 1), dtype=F32, device=g.device)
                slot, mode = param_grad_target(conv.weight)
                if bias_in_wgrad:
                    bslot, bmode = param_grad_target(conv.bias)
                    if bmode == 2:     # foreign .grad tensor on the bias: compute into the slot, commit adds it
                        bacc = 0
                    else:
                        bacc = 1 if bmode == 1 else 0
                    _C.check(lib.tok_conv_wgrad_bias(d, ptr(x.data), ptr(dy), ptr(slot), k, c, ptr(ws), ws_bytes,
                                                     1 if mode == 1 else 0, ptr(bslot), bacc, stream_ptr()),
                             'tok_conv_wgrad_bias')
                    commit_param_grad(conv.bias, bslot, bmode)
                else:
                    _C.check(lib.tok_conv_wgrad(d, ptr(x.data), ptr(dy), ptr(slot), k, c, ptr(ws), ws_bytes,
                                                1 if mode == 1 else 0, stream_ptr()), 'tok_conv_wgrad')
                commit_param_grad(conv.weight, slot, mode)
                return ws
            # LDS/MFMA-bound (3x3) and short-M weight gradients complement the HBM-bound main chain; the long-M pointwise ones
            # are HBM-bound themselves and only fight it for bandwidth
            side_ok = (WGRAD_SIDE_WHICH == 'all' or (r * s > 1 and WGRAD_SIDE_WHICH != '1x1') or
                       (m < WGRAD_SIDE_MAX_ROWS and (r * s == 1 or WGRAD_SIDE_WHICH != '1x1')) or
                       (r * s == 1 and _pointwise_wgrad_is_mfma_bound(k, c)))
            if WGRAD_SIDE_STREAM and side_ok and _side_for_tag(self.stream_tag, self.region) and g.is_cuda and self.region is not None \
                    and (SIDE_IN_GRAPH or not torch.cuda.is_current_stream_capturing()):
                # nothing on the main chain waits for dW: the weight gradient (LDS/MFMA-bound) runs on the side stream
                # beside the HBM-bound BatchNorm passes and the dgrad of the units below; joined at the end of the region
                with self.region.fork_side((x.data, dy), raw_event=apply_event):
                    self.region.keep_until_join(run_wgrad())
            elif self.region is not None and g.is_cuda:
                self.region.defer_wgrad(run_wgrad)     # (the closure keeps x and dy alive)
            else:
                run_wgrad()
        # a 3x3 weight gradient started BEFORE its unit's 3x3 data gradient runs beside it — two LDS/MFMA-bound kernels
        # sharing the LDS pipes; started AFTER it, it runs beside the HBM-bound BatchNorm / pointwise kernels that follow
        if w_need and not WGRAD_AFTER_DGRAD:
            launch_wgrad()
        if x_need:
            prod = x.node
            fuse = (isinstance(prod, _ConvBnActNode) and is_last_contribution(x) and prod.wants_fused_bwd_stats()
                    and prod.fused_partial is None and prod.fused_coef is None)
            # (a fused unit WITHOUT activation has no ReLU bits: its d(out) is dz itself and takes the plain path)
            mask_fuse = (isinstance(prod, _Unit3Node) and prod.relu and prod.mask is not None and is_last_contribution(x)
                         and prod.masked_partial is None)
            sub = x.grad_sub if getattr(self, 'sub_capable', False) else None
            tgt, acc = grad_target(x, sub_ok=sub is not None)
            if sub is not None:
                # d(x) = this data gradient + the parked gradient of x[:, ::2, ::2] (a strided projection shortcut): one
                # launch, d(x) written once; the epilogue variants of the plain call sites below
                x.grad_sub = None
                assert acc == 0
                rows = lib.tok_conv_dgrad_stat_rows(d)
                if mask_fuse:
                    partial = torch.empty((2, rows, x.cp), dtype=F32, device=g.device)
                    _C.check(lib.tok_conv_dgrad_subacc(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), ptr(sub), None, ptr(prod.mask),
                                                       ptr(partial), 1, st), 'tok_conv_dgrad_subacc')
                    prod.masked_partial = (partial, rows)
                elif fuse:
                    partial = torch.empty((2, rows, x.cp), dtype=F32, device=g.device)
                    _C.check(lib.tok_conv_dgrad_subacc(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), ptr(sub), ptr(prod.y),
                                                       ptr(prod.mask) if prod.relu else None, ptr(partial), 0, st),
                             'tok_conv_dgrad_subacc')
                    prod.fused_partial = (partial, rows)
                else:
                    _C.check(lib.tok_conv_dgrad_subacc(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), ptr(sub), None, None, None, 0,
                                                       st), 'tok_conv_dgrad_subacc')
            elif mask_fuse:
                # this dgrad completes the gradient of a fused unit-3 output: its epilogue stores dz = relu_mask * d(out)
                # (what that unit's backward and its shortcut both consume) and reduces sum(dz)
                rows = lib.tok_conv_dgrad_stat_rows(d)
                partial = torch.empty((2, rows, x.cp), dtype=F32, device=g.device)
                _C.check(lib.tok_conv_dgrad_maskstore(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), acc, ptr(prod.mask),
                                                      ptr(partial), st), 'tok_conv_dgrad_maskstore')
                prod.masked_partial = (partial, rows)
            elif fuse:
                # this dgrad completes d(x): its epilogue also reduces the BatchNorm-backward sums of the
                # unit that produced x (saves that unit a full pass over d(x) and y)
                rows = lib.tok_conv_dgrad_stat_rows(d)
                partial = torch.empty((2, rows, x.cp), dtype=F32, device=g.device)
                pbn = prod.bn
                pg, pb = pbn.weight.requires_grad, pbn.bias.requires_grad
                gs, gm = param_grad_target(pbn.weight) if pg else (None, 0)
                bs, bm = param_grad_target(pbn.bias) if pb else (None, 0)
                simple = FUSE_BN_FINALIZE and gm != 2 and bm != 2 and not (pg and pb and gm != bm)
                if simple:
                    # ... and folds them: dgamma, dbeta, apply coefficients of the producer (last workgroup per
                    # channel tile) — the producer's backward starts directly with its apply pass
                    pm = prod.y.numel() // prod.y.shape[-1]
                    coef = torch.empty((3, x.cp), dtype=F32, device=g.device)
                    fb = _C.BnFused(_ticket_counters(g.device), pm, pbn.num_features, 1 if (gm == 1 or bm == 1) else 0,
                                    0.0, 0.0, ptr(pbn.weight), None, None, None, None, ptr(prod.mean), ptr(prod.rstd),
                                    None, None, ptr(gs), ptr(bs), ptr(coef))
                    _C.check(lib.tok_conv_dgrad_bn(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), acc, ptr(prod.y),
                                                   ptr(prod.mask) if prod.relu else None, ptr(partial), fb, st),
                             'tok_conv_dgrad_bn')
                    prod.fused_coef = coef
                    if pg:
                        commit_param_grad(pbn.weight, gs, gm)
                    if pb:
                        commit_param_grad(pbn.bias, bs, bm)
                else:
                    _C.check(lib.tok_conv_dgrad_bnstats(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), acc, ptr(prod.y),
                                                        ptr(prod.mask) if prod.relu else None, ptr(partial), st),
                             'tok_conv_dgrad_bnstats')
                    prod.fused_partial = (partial, rows)
            else:
                _C.check(lib.tok_conv_dgrad(d, ptr(dy), ptr(self.pk.dgrad), ptr(tgt), acc, st), 'tok_conv_dgrad')
        if w_need and WGRAD_AFTER_DGRAD:
            launch_wgrad()



# ---- unit 3 of a bottleneck: 1x1 conv -> BatchNorm -> + shortcut -> ReLU without the pre-normalisation tensor ----------

FUSE_UNIT3 = os.environ.get('TOK_FUSE_UNIT3', '1') != '0'
# the side-stream fork / join of the weight gradients inside a hipGraph capture (cross-stream capture): experiment switch
SIDE_IN_GRAPH = os.environ.get('TOK_SIDE_IN_GRAPH', '0') == '1'
BIAS_IN_WGRAD = os.environ.get('TOK_BIAS_IN_WGRAD', '1') != '0'
COLSUM_IN_ACT = os.environ.get('TOK_COLSUM_IN_ACT', '1') != '0'
LAUNCH_EVENTS = os.environ.get('TOK_LAUNCH_EVENTS', '1') != '0'
DGRAD2 = os.environ.get('TOK_DGRAD2', '1') != '0'
# the fused unit trades ~27 tensor-units of HBM traffic for a handful of small launches (Gram matrix, two K x P x P products):
# it pays where the 4P-channel maps are large (ResNet-50 at batch 256: layers 1-2 and, marginally, 3)
UNIT3_MIN_ROWS = int(os.environ.get('TOK_UNIT3_MIN_ROWS', '100000'))   # measured: 0 -> 22.4, 40000 -> 22.0, 100000 -> 21.8, plain 23.2 ms/step


def _pointwise_desc(x: TTensor, k: int) -> _C.ConvDesc:
    n, h, w, cp = x.shape
    return _C.ConvDesc(n, h, w, cp, k, 1, 1, h, w, 1, 0, 1)


def _colsum_f32(lib, st, t: torch.Tensor, m: int, c: int) -> torch.Tensor:
    out = torch.empty(c, dtype=F32, device=t.device)
    if m > 4096:
        nrows = lib.tok_colsum_partial_rows(m, c)
        part = torch.empty((nrows, c), dtype=F32, device=t.device)
        _C.check(lib.tok_colsum_partial(ptr(t), m, c, ptr(part), st), 'tok_colsum_partial')
        _C.check(lib.tok_colsum_f32(ptr(part), nrows, c, ptr(out), 0, st), 'tok_colsum_f32')
    else:
        _C.check(lib.tok_colsum(ptr(t), m, c, c, ptr(out), 0, st), 'tok_colsum')
    return out


def _wgrad_f32(lib, st, d: _C.ConvDesc, x: torch.Tensor, dy: torch.Tensor, k: int, c: int, out: torch.Tensor = None,
               accumulate: int = 0) -> torch.Tensor:
    """out[k][c] (fp32) = dy^T x through the weight-gradient kernels (also used for the Gram matrix z^T z)."""
    if out is None:
        out = torch.empty((k, c), dtype=F32, device=x.device)
    ws_bytes = lib.tok_conv_wgrad_ws_bytes(d)
    ws = torch.empty(max(ws_bytes // 4, 1), dtype=F32, device=x.device)
    _C.check(lib.tok_conv_wgrad(d, ptr(x), ptr(dy), ptr(out), k, c, ptr(ws), ws_bytes, accumulate, st), 'tok_conv_wgrad')
    return out


class _Unit3Node(Node):
    """out = relu(bn(conv1x1(x)) + shortcut) with the BatchNorm statistics taken from the second moments of x and the whole
    backward written on x, dz = relu_mask * d(out) and small P x P / K x P matrices (csrc/unit3.hip).  The K-channel tensor
    between conv and BatchNorm (and its gradient) never exists."""
    needs_backward = True

    def __init__(self):
        self.x = self.out = self.shortcut = self.mask = None
        self.relu = True
        self.masked_partial = None      # (partial, rows): the launch that completed d(out) already stored dz and sum(dz)

    def release(self):
        self.x = self.out = self.shortcut = self.mask = self.masked_partial = None
        self.mean = self.rstd = self.wz = self.zsum = self.pk = None

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        out: TTensor = self.out
        g = out.grad
        if g is None:
            return
        conv, bn, d, x, sc = self.conv, self.bn, self.desc, self.x, self.shortcut
        kp, p = d.k, d.c
        m = d.n * d.p * d.q
        dev = g.device
        # 1. dz = relu_mask * d(out) (d(out) itself for a unit without activation) and the partial sums of dz
        if self.masked_partial is not None:
            partial, rows = self.masked_partial
            dz = g
        else:
            rows = lib.tok_bn_bwd_rows(m, kp)
            partial = torch.empty((2, rows, kp), dtype=F32, device=dev)
            dz = g if (out.grad_owned or not self.relu) else torch.empty_like(g)
            _C.check(lib.tok_relu_mask_reduce(ptr(g), ptr(self.mask) if self.relu else None, m, kp, ptr(dz), ptr(partial), st),
                     'tok_relu_mask_reduce')
        out.grad = None
        # 2. the shortcut receives dz itself; a projection shortcut of the same kind (conv1x1 + BatchNorm, no activation,
        #    nobody else reads its output) inherits the partial sums of dz as well
        if sc is not None and sc.requires_grad:
            if donate_grad(sc, dz):
                scn = sc.node
                if isinstance(scn, _Unit3Node) and not scn.relu and sc.uses == 1 and scn.masked_partial is None:
                    scn.masked_partial = (partial, rows)
            else:
                tgt, _ = grad_target(sc)
                tgt.add_(dz)
        w_need = conv.weight.requires_grad
        x_need = x.requires_grad
        g_need, b_need = bn.weight.requires_grad, bn.bias.requires_grad
        if not (w_need or x_need or g_need or b_need):
            return
        # 3. G = dz^T x   (the weight-gradient launch, on dz)
        G = _wgrad_f32(lib, st, d, x.data, dz, kp, p)
        # 4. dgamma / dbeta / dW, coefficients and the two operands of the data gradient
        gs, gm = param_grad_target(bn.weight) if g_need else (None, 0)
        bs, bm = param_grad_target(bn.bias) if b_need else (None, 0)
        ws_, wm = param_grad_target(conv.weight) if w_need else (torch.empty((kp, p), dtype=F32, device=dev), 0)
        mixed = g_need and b_need and (gm == 1) != (bm == 1)
        if mixed:     # one accumulates in its slot, the other does not: take both through temporaries
            gbuf, bbuf = torch.empty_like(gs), torch.empty_like(bs)
        else:
            gbuf, bbuf = gs, bs
        pacc = 1 if (not mixed and (gm == 1 or bm == 1)) else 0
        coef = torch.empty((3, kp), dtype=F32, device=dev)
        wa = torch.empty((p, kp), dtype=BF16, device=dev)
        wb = torch.empty((p, p), dtype=BF16, device=dev)
        cvec = torch.empty(p, dtype=F32, device=dev)
        scratch = torch.empty(lib.tok_bn3_bwd_prepare_ws_floats(p, kp), dtype=F32, device=dev)
        _C.check(lib.tok_bn3_bwd_prepare(ptr(G), ptr(conv.weight), ptr(self.wz), ptr(self.zsum), ptr(partial), rows, m, p,
                                         kp, ptr(bn.weight), ptr(self.mean), ptr(self.rstd), ptr(gbuf), ptr(bbuf), pacc,
                                         ptr(coef), ptr(ws_), 1 if wm == 1 else 0, ptr(wa), ptr(wb), ptr(cvec), ptr(scratch),
                                         st), 'tok_bn3_bwd_prepare')
        for need, prm, slot, mode, buf in ((g_need, bn.weight, gs, gm, gbuf), (b_need, bn.bias, bs, bm, bbuf)):
            if not need:
                continue
            if mixed:
                if mode == 1:
                    slot.add_(buf)
                else:
                    slot.copy_(buf)
            commit_param_grad(prm, slot, mode)
        if w_need:
            commit_param_grad(conv.weight, ws_, wm)
        if not x_need:
            return
        # 5. d(x) = dz wa + x wb + cvec  (+ the BatchNorm-backward sums of the unit that produced x)
        prod = x.node
        fuse = (isinstance(prod, _ConvBnActNode) and is_last_contribution(x) and prod.wants_fused_bwd_stats()
                and prod.fused_partial is None and prod.fused_coef is None)
        tgt, acc = grad_target(x)
        dpp = _pointwise_desc(x, p)
        if DGRAD2 and lib.tok_conv_dgrad2_ok(d, dpp):
            # both products in one launch of the ring kernel: d(x) is stored once
            part2 = None
            if fuse:
                rows2 = lib.tok_conv_dgrad_stat_rows(dpp)
                part2 = torch.empty((2, rows2, p), dtype=F32, device=dev)
            _C.check(lib.tok_conv_dgrad2(d, ptr(dz), ptr(wa), dpp, ptr(x.data), ptr(wb), ptr(cvec), ptr(tgt), acc,
                                         ptr(prod.y) if fuse else None, ptr(prod.mask) if (fuse and prod.relu) else None,
                                         ptr(part2), st), 'tok_conv_dgrad2')
            if fuse:
                prod.fused_partial = (part2, rows2)
            if self.region is not None:
                self.region.keep_until_join(dz, wa, wb, cvec, G, scratch)
            return
        _C.check(lib.tok_conv_dgrad(d, ptr(dz), ptr(wa), ptr(tgt), acc, st), 'tok_conv_dgrad')
        if fuse:
            rows2 = lib.tok_conv_dgrad_stat_rows(dpp)
            part2 = torch.empty((2, rows2, p), dtype=F32, device=dev)
            _C.check(lib.tok_conv_dgrad_bias(dpp, ptr(x.data), ptr(wb), ptr(cvec), ptr(tgt), 1, ptr(prod.y),
                                             ptr(prod.mask) if prod.relu else None, ptr(part2), st), 'tok_conv_dgrad_bias')
            prod.fused_partial = (part2, rows2)
        else:
            _C.check(lib.tok_conv_dgrad_bias(dpp, ptr(x.data), ptr(wb), ptr(cvec), ptr(tgt), 1, None, None, None, st),
                     'tok_conv_dgrad_bias')
        if self.region is not None:
            self.region.keep_until_join(dz, wa, wb, cvec, G, scratch)


def _unit3_forward(region: Region, x: TTensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, shortcut: Optional[TTensor], kp: int,
                   relu: bool) -> TTensor:
    lib, st = _C.lib(), stream_ptr()
    dev = x.data.device
    if bn.momentum is None:
        raise NotImplementedError('BatchNorm momentum=None (cumulative average)')
    d = _pointwise_desc(x, kp)
    p = x.cp
    m = d.n * d.p * d.q
    pk = get_packs(conv.weight, None, kp, 1, p, want_dgrad=False, refresh=True)
    # batch statistics of conv(x) from the second moments of x:  Z = x^T x  and  colsum(x)
    dz_ = _pointwise_desc(x, p)
    zz = _wgrad_f32(lib, st, dz_, x.data, x.data, p, p)
    if x.colsum_part is not None:
        part, nrows = x.colsum_part        # left behind by the activation pass that produced x
        zsum = torch.empty(p, dtype=F32, device=x.data.device)
        _C.check(lib.tok_colsum_f32(ptr(part), nrows, p, ptr(zsum), 0, st), 'tok_colsum_f32')
        x.colsum_part = None
    else:
        zsum = _colsum_f32(lib, st, x.data, m, p)
    mean, rstd, scale, shift = (torch.empty(kp, dtype=F32, device=dev) for _ in range(4))
    wz = torch.empty((kp, p), dtype=F32, device=dev)       # W Z: the statistics now, the weight gradient later
    track = bn.training and bn.track_running_stats and bn.running_mean is not None
    _C.check(lib.tok_bn_gram_finalize(ptr(zz), ptr(zsum), ptr(conv.weight), m, p, kp, ptr(bn.weight), ptr(bn.bias),
                                      ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                                      ptr(bn.num_batches_tracked) if track else None, float(bn.momentum), float(bn.eps),
                                      ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ptr(wz), st), 'tok_bn_gram_finalize')
    out_data = torch.empty((d.n, d.p, d.q, kp), dtype=BF16, device=dev)
    training = region.grad_mode and (conv.weight.requires_grad or x.requires_grad or bn.weight.requires_grad
                                     or (shortcut is not None and shortcut.requires_grad))
    mask = torch.empty((m, kp // 8), dtype=torch.uint8, device=dev) if (training and relu) else None
    _C.check(lib.tok_conv_fwd_bn_apply(d, ptr(x.data), ptr(pk.fwd), ptr(scale), ptr(shift),
                                       ptr(shortcut.data) if shortcut is not None else None, int(relu), ptr(out_data),
                                       ptr(mask), st), 'tok_conv_fwd_bn_apply')
    out = TTensor(out_data, kp, requires_grad=bool(training))
    if training:
        node = _Unit3Node()
        node.x, node.out, node.shortcut, node.mask, node.relu = x, out, shortcut, mask, relu
        node.conv, node.bn, node.desc, node.pk = conv, bn, d, pk
        node.mean, node.rstd, node.wz, node.zsum = mean, rstd, wz, zsum
        out.node = node
        if x.requires_grad:
            x.uses += 1
        if shortcut is not None and shortcut.requires_grad:
            shortcut.uses += 1
        region.add(node)
    return out


FUSE_STEM_POOL = os.environ.get('TOK_FUSE_STEM_POOL', '1') != '0'
STEM_POOLED_STATS = os.environ.get('TOK_STEM_POOLED_STATS', '1') != '0'   # BatchNorm-backward sums of the fused stem in the pooled domain


def conv_bn_act(region: Region, x: TTensor, conv: nn.Module, bn: Optional[nn.BatchNorm2d] = None,
                relu: bool = False, shortcut: Optional[TTensor] = None, pool: bool = False, defer_apply: bool = False) -> TTensor:
    """out = act(bn(conv(x)) (+ shortcut)).  `conv` is an nn.Conv2d or nn.Linear used purely as
    the parameter container (state_dict names stay those of the reference).
    pool=True (BatchNorm + ReLU, no shortcut): out = maxpool3x3/s2/p1(act(bn(conv(x)))) with the activated map never
    stored (the ResNet stem when only the pooled map is consumed).
    defer_apply=True (BatchNorm, no activation, no shortcut — the last unit of an HRNet fuse path): the apply pass is NOT run;
    the returned tensor holds the RAW convolution output and carries `.affine = (scale, shift)` for a consumer that applies
    them itself (resample.fuse_sum_relu: the write and the read of the normalised term are one fma there).  Its backward is
    the unit's usual one (it never reads the normalised values of a unit without activation)."""
    if pool and (bn is None or not relu or shortcut is not None or x.data.dim() != 4):
        raise ValueError('conv_bn_act(pool=True): BatchNorm + ReLU on a 4-D input, no shortcut')
    await_ready(x, shortcut)
    lib, st = _C.lib(), stream_ptr()
    if isinstance(conv, nn.Linear):
        r = s = 1
        stride, pad = 1, 0
        k_real = conv.out_features
    else:
        _check_conv(conv)
        r, s = conv.kernel_size
        stride, pad = conv.stride[0], conv.padding[0]
        k_real = conv.out_channels
    if (SUBSAMPLE_S2 and r == 1 and s == 1 and stride == 2 and pad == 0 and x.data.dim() == 4 and not pool
            and x.rows() >= SUBSAMPLE_MIN_ROWS):
        # conv1x1/stride2(x) == conv1x1(x[:, ::2, ::2]): one strided copy, then a pointwise layer in all three passes
        x = subsample2(region, x)
        stride = 1
    x4 = x
    kp = pad8(k_real)
    if bn is not None and bn.num_features != k_real:
        raise ValueError(f'BatchNorm num_features {bn.num_features} != conv output channels {k_real}')
    if (FUSE_UNIT3 and bn is not None and (relu == (shortcut is not None)) and not pool and isinstance(conv, nn.Conv2d)
            and r == 1 and s == 1 and stride == 1 and pad == 0 and conv.bias is None and x.data.dim() == 4
            and x.c == x.cp and kp == k_real and x.cp <= 1024 and (shortcut is None or shortcut.cp == kp)
            and x.rows() >= UNIT3_MIN_ROWS and kp >= 2 * x.cp
            and (bn.training or bn.running_mean is None) and conv.weight.permute(0, 2, 3, 1).is_contiguous()):
        # the residual unit of a bottleneck (conv3 + bn3 + shortcut + ReLU) and its stride-1 projection shortcut (conv + bn):
        # normalise (add, activate) in the GEMM epilogue — the wide pre-BatchNorm tensor is never stored
        return _unit3_forward(region, x, conv, bn, shortcut, kp, relu)
    d = _conv_desc(x4, kp, r, s, stride, pad)
    training = region.grad_mode and (conv.weight.requires_grad or x.requires_grad or
                                     (bn is not None and bn.weight.requires_grad))
    pk = get_packs(conv.weight, conv.bias, kp, d.s_pad, x4.cp, want_dgrad=region.grad_mode and x.requires_grad,
                   refresh=True)
    dev = x.data.device
    y = torch.empty((d.n, d.p, d.q, kp), dtype=BF16, device=dev)
    m = d.n * d.p * d.q
    batch_stats = bn is not None and (bn.training or bn.running_mean is None)
    stats = None
    if batch_stats:
        rows = lib.tok_conv_fwd_stat_rows(d)
        stats = torch.empty((2, rows, kp), dtype=F32, device=dev)
    fused_fin = batch_stats and pk.bias is None and FUSE_BN_FINALIZE
    node = _ConvBnActNode()
    if fused_fin:
        # the conv launch also folds its statistics rows (last workgroup per channel tile): no finalize launch
        if bn.momentum is None:
            raise NotImplementedError('BatchNorm momentum=None (cumulative average)')
        scale, shift, mean, rstd = (torch.empty(kp, dtype=F32, device=dev) for _ in range(4))
        track = bn.training and bn.track_running_stats and bn.running_mean is not None
        fb = _C.BnFused(_ticket_counters(dev), m, bn.num_features, 0, float(bn.momentum), float(bn.eps), ptr(bn.weight),
                        ptr(bn.bias), ptr(bn.running_mean) if track else None, ptr(bn.running_var) if track else None,
                        ptr(bn.num_batches_tracked) if track else None, ptr(mean), ptr(rstd), ptr(scale), ptr(shift),
                        None, None, None)
        _C.check(lib.tok_conv_fwd_bn(d, ptr(x4.data), ptr(pk.fwd), ptr(y), ptr(stats), fb, st), 'tok_conv_fwd_bn')
    else:
        _C.check(lib.tok_conv_fwd(d, ptr(x4.data), ptr(pk.fwd), ptr(pk.bias), ptr(y), ptr(stats), st), 'tok_conv_fwd')

    if bn is not None:
        if not fused_fin:
            # one allocation for the per-channel vectors (an allocator call costs the launch thread 2-3 us; HRNet-W48 makes 307
            # of these units per step)
            vec = torch.empty((4, kp), dtype=F32, device=dev)
            scale, shift = vec[0], vec[1]
            mean = rstd = None
        if batch_stats and not fused_fin:
            if bn.momentum is None:
                raise NotImplementedError('BatchNorm momentum=None (cumulative average)')
            mean, rstd = vec[2], vec[3]
            track = bn.training and bn.track_running_stats and bn.running_mean is not None
            _C.check(lib.tok_bn_finalize(ptr(stats), rows, m, kp, bn.num_features, ptr(bn.weight), ptr(bn.bias),
                                         ptr(bn.running_mean) if track else None,
                                         ptr(bn.running_var) if track else None,
                                         ptr(bn.num_batches_tracked) if track else None,
                                         float(bn.momentum), float(bn.eps), ptr(mean), ptr(rstd),
                                         ptr(scale), ptr(shift), st), 'tok_bn_finalize')
        elif not batch_stats:
            _C.check(lib.tok_bn_eval_coeffs(ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                                            ptr(bn.running_var), float(bn.eps), kp, bn.num_features, ptr(scale), ptr(shift), st),
                     'tok_bn_eval_coeffs')
        mask = None
        deferred = bool(defer_apply and not relu and shortcut is None and not pool and x.data.dim() == 4 and not fused_fin)
        if deferred:
            out_data = y
            cs_part = None
        elif pool:
            p2, q2 = (d.p + 2 - 3) // 2 + 1, (d.q + 2 - 3) // 2 + 1
            out_data = torch.empty((d.n, p2, q2, kp), dtype=BF16, device=dev)
            node.pool = torch.empty((d.n, p2, q2, kp), dtype=torch.uint8, device=dev)
            if STEM_POOLED_STATS and region.grad_mode and batch_stats:
                node.ypool = torch.empty_like(out_data)
            _C.check(lib.tok_bn_relu_maxpool_fwd(ptr(y), ptr(scale), ptr(shift), d.n, d.p, d.q, kp, ptr(out_data),
                                                 ptr(node.pool), ptr(node.ypool), st), 'tok_bn_relu_maxpool_fwd')
        else:
            out_data = torch.empty_like(y)
            if relu and region.grad_mode and batch_stats:
                mask = torch.empty((m, kp // 8), dtype=torch.uint8, device=dev)   # ReLU bits for the backward pass
            # a 3x3 unit of a bottleneck feeds the fused residual unit, which wants colsum(z) of its input: the activation pass
            # has z in registers (saves that unit a stand-alone pass over z)
            want_cs = (FUSE_UNIT3 and COLSUM_IN_ACT and r == 3 and relu and shortcut is None and m >= UNIT3_MIN_ROWS
                       and kp == k_real and kp <= 1024 and region.grad_mode and batch_stats)
            cs_part = None
            if want_cs:
                cs_rows = lib.tok_bn_act_fwd_colsum_rows(m, kp)
                cs_part = torch.empty((cs_rows, kp), dtype=F32, device=dev)
            if want_cs:
                _C.check(lib.tok_bn_act_fwd_colsum(ptr(y), ptr(scale), ptr(shift), None, int(relu), ptr(out_data), ptr(mask),
                                                   m, kp, ptr(cs_part), st), 'tok_bn_act_fwd_colsum')
            else:
                _C.check(lib.tok_bn_act_fwd(ptr(y), ptr(scale), ptr(shift),
                                            ptr(shortcut.data) if shortcut is not None else None,
                                            int(relu), ptr(out_data), ptr(mask), m, kp, st), 'tok_bn_act_fwd')
        node.mask = mask
        node.mean, node.rstd, node.scale, node.shift = mean, rstd, scale, shift
    else:
        if relu or shortcut is not None:
            raise NotImplementedError('relu / shortcut without BatchNorm')
        out_data = y
    if x.data.dim() == 2:
        y = y.view(d.n, kp)
        out_data = out_data.view(d.n, kp)

    req = bool(training or (shortcut is not None and shortcut.requires_grad and region.grad_mode))
    out = TTensor(out_data, k_real, requires_grad=req)
    if bn is not None and deferred:
        out.affine = (scale, shift)
    if bn is not None and not pool and cs_part is not None:
        out.colsum_part = (cs_part, cs_rows)
    if req:
        node.x, node.out, node.shortcut, node.y = x, out, shortcut, y
        node.conv, node.bn, node.desc, node.pk = conv, bn, d, pk
        node.relu, node.batch_stats = relu, batch_stats
        out.node = node
        node.sub_capable = False
        if x.requires_grad:
            x.uses += 1
            if (r == 1 and s == 1 and stride == 1 and pad == 0 and x.data.dim() == 4
                    and lib.tok_conv_dgrad_subacc_ok(d)):
                node.sub_capable = True      # this unit's data gradient can absorb a half-resolution contribution to d(x)
                x.sub_closers += 1
        if shortcut is not None and shortcut.requires_grad:
            shortcut.uses += 1
        region.add(node)
    return out


def linear(region: Region, x: TTensor, fc: nn.Linear) -> TTensor:
    """y = x W^T + b on the MFMA conv kernel (1x1, h = w = 1)."""
    return conv_bn_act(region, x, fc, None, False, None)


# ---- max pool ----------------------------------------------------------------------------------------

class _MaxPoolNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        n, h, w, c = self.x.shape
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_maxpool3x3s2_bwd(ptr(g), ptr(self.argmax), ptr(tgt), acc, n, h, w, c, stream_ptr()),
                 'tok_maxpool3x3s2_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = self.argmax = None


def max_pool_3x3_s2(region: Region, x: TTensor) -> TTensor:
    n, h, w, c = x.shape
    p, q = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = torch.empty((n, p, q, c), dtype=BF16, device=x.data.device)
    argmax = torch.empty((n, p, q, c), dtype=torch.uint8, device=x.data.device)
    _C.check(_C.lib().tok_maxpool3x3s2_fwd(ptr(x.data), ptr(y), ptr(argmax), n, h, w, c, stream_ptr()),
             'tok_maxpool3x3s2_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _MaxPoolNode()
        node.x, node.out, node.argmax = x, out, argmax
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- 2x2 average pool (avg_down shortcuts) ---------------------------------------------------------------------

class _Subsample2Node(Node):
    """x -> x[:, ::2, ::2]: the input of a 1x1 / stride-2 projection as a dense tensor.  Its backward does not touch d(x)
    when exactly one more contribution is due and that consumer's data gradient can absorb the half-resolution gradient
    in its epilogue (tok_conv_dgrad_subacc): the buffer is parked on `x.grad_sub`."""
    needs_backward = True

    def backward(self):
        g = self.out.grad
        x = self.x
        if g is None or not x.requires_grad:
            return
        self.out.grad = None
        if (x.grad is None and x.grad_sub is None and x.uses == 2 and x.arrived == 0 and x.sub_closers == 1
                and not getattr(_core_ms, 'active', False)):
            x.grad_sub = g
            x.arrived += 1
            return
        n, h, w, c = x.shape
        tgt, acc = grad_target(x)
        _C.check(_C.lib().tok_subsample2_bwd(ptr(g), n, h, w, c, ptr(tgt), acc, stream_ptr()), 'tok_subsample2_bwd')

    def release(self):
        self.x = self.out = None


# 1x1 / stride-2 / no-padding convolutions run as pointwise layers on the subsampled input (forward, weight gradient, data
# gradient on the streaming kernels; the unit-3 fusion applies to the projection + BatchNorm).  Measured on ResNet-50 B=256:
# see DESIGN.md section 8.
SUBSAMPLE_S2 = os.environ.get('TOK_SUBSAMPLE_S2', '1') != '0'
# rows of the full-resolution input (round 6: 100 000 -> 40 000 puts ResNet-50's layer4.0 projection at batch 256 — 50 176 rows, the
# last stride-2 1x1 data gradient on the parity-class kernel — on the subsampled plan too: 17.42 vs 17.50 ms/step, two same-box rounds)
SUBSAMPLE_MIN_ROWS = int(os.environ.get('TOK_SUBSAMPLE_MIN_ROWS', '40000'))


def subsample2(region: Region, x: TTensor) -> TTensor:
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=BF16, device=x.data.device)
    _C.check(_C.lib().tok_subsample2_fwd(ptr(x.data), n, h, w, c, ptr(y), stream_ptr()), 'tok_subsample2_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _Subsample2Node()
        node.x, node.out = x, out
        out.node = node
        x.uses += 1
        region.add(node)
    return out


class _AvgPool2Node(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        n, h, w, c = self.x.shape
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_avgpool2x2_bwd(ptr(g), ptr(tgt), acc, n, h, w, c, stream_ptr()), 'tok_avgpool2x2_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def avg_pool_2x2(region: Region, x: TTensor) -> TTensor:
    """AvgPool2d(2, stride 2, ceil_mode=True, count_include_pad=False) ([timm] downsample_avg)."""
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=BF16, device=x.data.device)
    _C.check(_C.lib().tok_avgpool2x2_fwd(ptr(x.data), ptr(y), n, h, w, c, stream_ptr()), 'tok_avgpool2x2_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _AvgPool2Node()
        node.x, node.out = x, out
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- global average pool --------------------------------------------------------------------------------

class _GapNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        n, h, w, c = self.x.shape
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_gap_bwd(ptr(g), ptr(tgt), acc, n, h * w, c, stream_ptr()), 'tok_gap_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def global_avg_pool(region: Region, x: TTensor) -> TTensor:
    n, h, w, c = x.shape
    y = torch.empty((n, c), dtype=BF16, device=x.data.device)
    _C.check(_C.lib().tok_gap_fwd(ptr(x.data), ptr(y), n, h * w, c, stream_ptr()), 'tok_gap_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _GapNode()
        node.x, node.out = x, out
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- global max / avgmax / catavgmax pool ------------------------------------------------------------------

POOL_MODES = {'max': 1, 'avgmax': 2, 'catavgmax': 3}


class _GlobalPoolNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        n, h, w, c = self.x.shape
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_global_pool_bwd(ptr(g), ptr(self.argmax), ptr(tgt), acc, n, h * w, c, g.shape[-1], self.mode,
                                              stream_ptr()), 'tok_global_pool_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = self.argmax = None


def global_pool(region: Region, x: TTensor, pool_type: str) -> TTensor:
    """SelectAdaptivePool2d(1, pool_type, flatten=True) ([timm], reference pooling.py:7-12): 'avg', 'max',
    'avgmax' = 0.5 * (avg + max), 'catavgmax' = cat(avg, max) along channels."""
    if pool_type == 'avg':
        return global_avg_pool(region, x)
    mode = POOL_MODES[pool_type]
    n, h, w, c = x.shape
    if mode == 3 and x.c != c:
        raise NotImplementedError("torchok_amd Pooling 'catavgmax': channel counts that are multiples of 8")
    cout = 2 * c if mode == 3 else c
    y = torch.empty((n, cout), dtype=BF16, device=x.data.device)
    argmax = torch.empty((n, c), dtype=torch.int32, device=x.data.device)
    _C.check(_C.lib().tok_global_pool_fwd(ptr(x.data), ptr(y), ptr(argmax), n, h * w, c, cout, mode, stream_ptr()),
             'tok_global_pool_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, 2 * x.c if mode == 3 else x.c, requires_grad=req)
    if req:
        node = _GlobalPoolNode()
        node.x, node.out, node.argmax, node.mode = x, out, argmax, mode
        out.node = node
        x.uses += 1
        region.add(node)
    return out
