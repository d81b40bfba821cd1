"""Engine units of the token-major (SwinV2) rows.  Tokens are rows of a bf16 [B*H*W][C] matrix.

Unit                 replaces ([timm 0.6.13] swin_transformer_v2 through torchok/models/backbones/swin.py)
-------------------  -------------------------------------------------------------------------------------
linear_op            F.linear (qkv with cat(q_bias, k_bias, v_bias); proj; Mlp.fc1/fc2; PatchMerging.reduction;
                     cpb_mlp) on the MFMA conv kernels (1x1, h = w = 1)
layer_norm           nn.LayerNorm, optionally fused with the res-post-norm residual  x + drop_path(norm(.))
activation           GELU (Mlp) / ReLU (cpb_mlp)
cpb_bias             16 * sigmoid(cpb_mlp(relative_coords_table))[relative_position_index]
window_attention     WindowAttention.forward between qkv and proj, with roll / window_partition / window_reverse
                     folded into the kernel's token addressing
patch_merge          PatchMerging's strided 2x2 gather + cat
reshape              .view between (B, H, W, C) maps and (B*H*W, C) token rows (zero copy)
"""
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from .. import _C
from .core import (BF16, Node, Region, TTensor, await_mark, await_ready, commit_param_grad, donate_grad, grad_target,
                   pad8, param_grad_target, ptr, stream_ptr, written_mark)
from .functional import BIAS_IN_WGRAD, _krsc, get_packs

F32 = torch.float32


def _use_side(node, g: torch.Tensor) -> bool:
    from . import functional as EF
    return bool(EF.WGRAD_SIDE_STREAM and g.is_cuda and node.region is not None
                and not torch.cuda.is_current_stream_capturing())


def _rows(t: TTensor) -> int:
    return t.data.numel() // t.data.shape[-1]


# ---- reshape (zero copy) --------------------------------------------------------------------------------
class _ReshapeNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        gv = g.view(self.x.data.shape)
        if not donate_grad(self.x, gv):
            tgt, _ = grad_target(self.x)
            _C.check(_C.lib().tok_act_bwd(2, ptr(gv), ptr(gv), ptr(tgt), 1, gv.numel(), stream_ptr()), 'tok_act_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def reshape(region: Region, x: TTensor, shape: Sequence[int]) -> TTensor:
    out = TTensor(x.data.view(*shape), x.c, requires_grad=x.requires_grad and region.grad_mode)
    if out.requires_grad:
        node = _ReshapeNode()
        node.x, node.out = x, out
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- linear --------------------------------------------------------------------------------------------------
_LIN_PLANS = {}


def _linear_plan(lib, d):
    """(bias gradient rides the weight-gradient kernel?, workspace bytes with / without it) of a token-matrix geometry."""
    key = (d.n, d.c, d.k, id(lib))
    p = _LIN_PLANS.get(key)
    if p is None:
        p = (bool(lib.tok_conv_wgrad_bias_ok(d)), int(lib.tok_conv_wgrad_bias_ws_bytes(d)), int(lib.tok_conv_wgrad_ws_bytes(d)))
        _LIN_PLANS[key] = p
    return p


class _LinearNode(Node):
    needs_backward = True
    mlp_first = None        # fc2 of a fused Mlp: the node of its fc1 (tok_mlp_bwd_dx covers both data gradients)
    dx_done = False         # fc1 of a fused Mlp: its data gradient was produced by fc2's backward launch

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        x, w, d = self.x, self.weight, self.desc
        m, kp = g.shape
        side = _use_side(self, g)

        def scatter_bias(tmp):
            for p, start in self.bias_sinks:
                if not p.requires_grad:
                    continue
                slot, mode = param_grad_target(p)
                seg = tmp[start:start + p.numel()]
                if mode == 1:
                    slot.add_(seg)
                else:
                    slot.copy_(seg)
                commit_param_grad(p, slot, mode)
        # the column sums of g (bias gradients) come out of the weight-gradient kernel where it serves the layer
        # (plan facts of the geometry are cached: two library calls less per layer and step on the launch thread)
        plan = _linear_plan(lib, d)
        bias_in_wgrad = bool(self.bias_sinks and w.requires_grad and BIAS_IN_WGRAD and plan[0])
        if self.bias_sinks and not bias_in_wgrad:
            def run_bias():
                tmp = torch.empty(kp, dtype=F32, device=g.device)
                nrows = lib.tok_colsum_partial_rows(m, kp)
                part = torch.empty((nrows, kp), dtype=F32, device=g.device)
                st_ = stream_ptr()
                _C.check(lib.tok_colsum_partial(ptr(g), m, kp, ptr(part), st_), 'tok_colsum_partial')
                _C.check(lib.tok_colsum_f32(ptr(part), nrows, kp, ptr(tmp), 0, st_), 'tok_colsum_f32')
                scatter_bias(tmp)
                return tmp, part
            bias_fn = run_bias
        else:
            bias_fn = None
        wgrad_fn = None
        if w.requires_grad:
            k, r, s, c = _krsc(w)
            ws_bytes = plan[1] if bias_in_wgrad else plan[2]
            # one bias parameter spanning all output features (fc1 / fc2 / proj): the kernel writes its gradient slot itself
            sole = None
            if bias_in_wgrad and len(self.bias_sinks) == 1:
                bp, bstart = self.bias_sinks[0]
                if bstart == 0 and bp.numel() == k and bp.requires_grad:
                    sole = bp

            def run_wgrad():
                ws = torch.empty(max(ws_bytes // 4, 1), dtype=F32, device=g.device)
                slot, mode = param_grad_target(w)
                if bias_in_wgrad and sole is not None:
                    bslot, bmode = param_grad_target(sole)
                    _C.check(lib.tok_conv_wgrad_bias(d, ptr(x.data), ptr(g), ptr(slot), k, c, ptr(ws), ws_bytes,
                                                     1 if mode == 1 else 0, ptr(bslot), 1 if bmode == 1 else 0, stream_ptr()),
                             'tok_conv_wgrad_bias')
                    commit_param_grad(sole, bslot, bmode)
                elif bias_in_wgrad:
                    tmp = torch.empty(kp, dtype=F32, device=g.device)
                    _C.check(lib.tok_conv_wgrad_bias(d, ptr(x.data), ptr(g), ptr(slot), k, c, ptr(ws), ws_bytes,
                                                     1 if mode == 1 else 0, ptr(tmp), 0, stream_ptr()), 'tok_conv_wgrad_bias')
                    scatter_bias(tmp)
                else:
                    _C.check(lib.tok_conv_wgrad(d, ptr(x.data), ptr(g), ptr(slot), k, c, ptr(ws), ws_bytes,
                                                1 if mode == 1 else 0, stream_ptr()), 'tok_conv_wgrad')
                commit_param_grad(w, slot, mode)
                return ws
            wgrad_fn = run_wgrad

        def param_grads(event=None):
            if bias_fn is not None:
                if side:       # parameter gradients only: off the main chain, beside it (see functional.py)
                    with self.region.fork_side((g,), event=event):
                        self.region.keep_until_join(*bias_fn())
                else:
                    bias_fn()
            if wgrad_fn is not None:
                if side:
                    with self.region.fork_side((x.data, g), event=event):       # dW beside the main chain (see functional.py)
                        self.region.keep_until_join(wgrad_fn())
                else:
                    wgrad_fn()
        # the data gradient is the main chain: with DGRAD_FIRST the host enqueues it before the side-stream work (fork, scratch
        # allocation, weight-gradient + bias launches, hooks) — the side kernels still start where they used to (the fork event
        # is recorded up front).  Measured on SwinV2-T B=256: the main queue idled 100-190 us per block behind that host work.
        early = DGRAD_FIRST and side and x.requires_grad
        ev = self.region.mark_side() if early else None
        if not early:
            param_grads()
        if x.requires_grad and self.dx_done:
            pass            # fc1 of a fused Mlp: fc2's backward launch already sent this gradient on (tok_mlp_bwd_dx)
        elif x.requires_grad:
            an = x.node
            first = self.mlp_first
            if (FUSE_ACT and isinstance(an, _ActNode) and an.out is x and an.kind in (RELU, GELU) and x.uses == 1 and
                    x.grad is None and an.x.requires_grad and an.x.uses == 1 and an.x.grad is None):
                if (first is not None and FUSE_MLP_BWD and an.kind == GELU and an.x is first.out and first.x.requires_grad
                        and first.pk.dgrad is not None):
                    # fc2 of a fused Mlp: d(pre) = (dy W2) * GELU'(pre) stays in registers and goes straight through W1 to the
                    # Mlp's input; the d(pre) rows are written only if fc1's parameter gradients will read them
                    w1, sinks = first.weight, first.bias_sinks
                    need_dpre = w1.requires_grad or any(p.requires_grad for p, _ in sinks)
                    tgt = grad_target(an.x)[0] if need_dpre else None
                    tgt_x, acc_x = grad_target(first.x)
                    _C.check(lib.tok_mlp_bwd_dx(ptr(g), ptr(self.pk.dgrad), ptr(an.x.data), ptr(first.pk.dgrad), ptr(tgt_x), acc_x,
                                                ptr(tgt), m, d.k, d.c, st), 'tok_mlp_bwd_dx')
                    first.dx_done = True
                else:
                    # this layer consumes act(h) and nothing else does: its dgrad epilogue multiplies by act'(h) and writes
                    # d(h) — the gradient of act(h) is never materialised, the activation node finds nothing to do
                    tgt, _ = grad_target(an.x)
                    _C.check(lib.tok_conv_dgrad_act(d, ptr(g), ptr(self.pk.dgrad), ptr(an.x.data), an.kind, ptr(tgt), st),
                             'tok_conv_dgrad_act')
            else:
                tgt, acc = grad_target(x)
                _C.check(lib.tok_conv_dgrad(d, ptr(g), ptr(self.pk.dgrad), ptr(tgt), acc, st), 'tok_conv_dgrad')
        if early:
            param_grads(ev)
        self.out.grad = None

    def release(self):
        self.x = self.out = self.pk = self.mlp_first = None


# GELU / ReLU in the epilogues of the GEMMs around it (tok_conv_fwd_act / tok_conv_dgrad_act).  Bit-identical to the separate
# launches.  Round 2 measured it NEUTRAL with libm's erff / expf (SwinV2-T B=256: 29.0 vs 29.0 ms/step — ~55 VALU per element
# stretched the epilogue by what the deleted passes cost); with the 14-instruction erf of tok_common.h (round 3) the fused form
# wins: SwinV2-T 24.87 -> 24.59, DaViT-T 24.14 -> 23.95 ms/step, and 12 activation-sized tensor passes per block leave the
# step's HBM traffic.  On by default; TOK_FUSE_ACT=0 restores the separate launches.
FUSE_ACT = os.environ.get('TOK_FUSE_ACT', '1') == '1'
DGRAD_FIRST = os.environ.get('TOK_DGRAD_FIRST', '0') == '1'   # measured neutral (SwinV2-T 24.98 vs 24.97, DaViT-T 24.33 vs 24.20 ms): off


def linear_op(region: Region, x: TTensor, weight: nn.Parameter, bias_vec: Optional[torch.Tensor] = None,
              bias_sinks: Sequence[Tuple[nn.Parameter, int]] = (), act: Optional[int] = None):
    """y = x W^T + bias_vec.  `bias_vec` is an fp32 vector of the (padded) output width; `bias_sinks` lists the
    parameters it was assembled from as (param, start column) — their gradients are column sums of dy.
    act = RELU / GELU: returns act(y) instead, produced by the same launch (y is kept for the backward)."""
    lib, st = _C.lib(), stream_ptr()
    k, c = weight.shape
    n, cp = x.shape
    if pad8(c) != cp:
        raise ValueError(f'linear_op: input width {cp} does not match in_features {c}')
    kp = pad8(k)
    need_dx = region.grad_mode and x.requires_grad
    pk = get_packs(weight, None, kp, 1, cp, want_dgrad=need_dx, refresh=True)
    if bias_vec is not None and bias_vec.shape[0] != kp:
        b = torch.zeros(kp, dtype=F32, device=bias_vec.device)
        b[:bias_vec.shape[0]] = bias_vec
        bias_vec = b
    d = _C.ConvDesc(n, 1, 1, cp, kp, 1, 1, 1, 1, 1, 0, 1)
    y = torch.empty((n, kp), dtype=BF16, device=x.data.device)
    y_act = None
    if act is not None and FUSE_ACT:
        y_act = torch.empty_like(y)
        _C.check(lib.tok_conv_fwd_act(d, ptr(x.data), ptr(pk.fwd), ptr(bias_vec), ptr(y), ptr(y_act), act, st),
                 'tok_conv_fwd_act')
    else:
        _C.check(lib.tok_conv_fwd(d, ptr(x.data), ptr(pk.fwd), ptr(bias_vec), ptr(y), None, st), 'tok_conv_fwd')
    out = _record_linear(region, x, weight, bias_sinks, y, d, pk)
    if act is None:
        return out
    return activation(region, out, act, precomputed=y_act)


def _record_linear(region: Region, x: TTensor, weight: nn.Parameter, bias_sinks, y: torch.Tensor, d, pk) -> TTensor:
    """The tape entry of y = x W^T + b, whoever launched it."""
    k = weight.shape[0]
    req = region.grad_mode and (x.requires_grad or weight.requires_grad or any(p.requires_grad for p, _ in bias_sinks))
    out = TTensor(y, k, requires_grad=req)
    if req:
        node = _LinearNode()
        node.x, node.out, node.weight, node.desc, node.pk = x, out, weight, d, pk
        node.bias_sinks = list(bias_sinks)
        out.node = node
        if x.requires_grad:
            x.uses += 1
        region.add(node)
    return out


def linear_module(region: Region, x: TTensor, fc: nn.Linear, act: Optional[int] = None) -> TTensor:
    if fc.bias is None:
        return linear_op(region, x, fc.weight, act=act)
    return linear_op(region, x, fc.weight, fc.bias.detach(), [(fc.bias, 0)], act=act)


# The whole Mlp from one launch (csrc/mlp_fused.hip): fc2 consumes GELU(fc1(x)) out of registers; in training the bf16
# pre-activation and activation rows are still written (the backward GEMMs read them), so the tape is the one the separate
# launches record and every tensor on it has the same bits.  Measured per call on the SwinV2-T B=256 shapes (fused+saved vs
# fc1+GELU launch + fc2 launch): C=96 366 vs 543 us, C=192 269 vs 368, C=384 203 vs 239.  TOK_FUSE_MLP=0: separate launches.
FUSE_MLP = os.environ.get('TOK_FUSE_MLP', '1') == '1'
# ... and its backward to the input from one launch too (tok_mlp_bwd_dx replaces tok_conv_dgrad_act + tok_conv_dgrad; the d(pre)
# rows are still written for fc1's weight gradient).  Per call: C=96 360 vs 600 us, C=192 267 vs 373, C=384 216 vs 242.
FUSE_MLP_BWD = os.environ.get('TOK_FUSE_MLP_BWD', '1') == '1'


def mlp_module(region: Region, x: TTensor, fc1: nn.Linear, fc2: nn.Linear) -> TTensor:
    """fc2(GELU(fc1(x))) — [timm 0.6.13] models/layers/mlp.py: Mlp.forward with drop = 0."""
    lib = _C.lib()
    n, cp = x.shape
    c, hid = fc1.in_features, fc1.out_features
    served = (FUSE_MLP and FUSE_ACT and fc1.bias is not None and fc2.bias is not None and cp == c
              and fc2.in_features == hid and fc2.out_features == c and bool(lib.tok_mlp_serves(n, c, hid)))
    if not served:
        h = linear_module(region, x, fc1, act=GELU)      # fc1 + GELU: one launch; fc2's dgrad applies GELU'
        return linear_module(region, h, fc2)
    st = stream_ptr()
    params = (fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    train = region.grad_mode and (x.requires_grad or any(p.requires_grad for p in params))
    pk1 = get_packs(fc1.weight, None, hid, 1, cp, want_dgrad=region.grad_mode and x.requires_grad, refresh=True)
    pk2 = get_packs(fc2.weight, None, c, 1, hid, want_dgrad=train, refresh=True)
    dev = x.data.device
    y = torch.empty((n, c), dtype=BF16, device=dev)
    pre = act = None
    if train:
        pre = torch.empty((n, hid), dtype=BF16, device=dev)
        act = torch.empty_like(pre)
    _C.check(lib.tok_mlp_fwd(ptr(x.data), ptr(pk1.fwd), ptr(fc1.bias.detach()), ptr(pk2.fwd), ptr(fc2.bias.detach()), ptr(y),
                             ptr(pre), ptr(act), n, c, hid, st), 'tok_mlp_fwd')
    if not train:
        return TTensor(y, c, requires_grad=False)
    d1 = _C.ConvDesc(n, 1, 1, cp, hid, 1, 1, 1, 1, 1, 0, 1)
    d2 = _C.ConvDesc(n, 1, 1, hid, c, 1, 1, 1, 1, 1, 0, 1)
    h_pre = _record_linear(region, x, fc1.weight, [(fc1.bias, 0)], pre, d1, pk1)
    h = activation(region, h_pre, GELU, precomputed=act)
    out = _record_linear(region, h, fc2.weight, [(fc2.bias, 0)], y, d2, pk2)
    if out.node is not None and h_pre.node is not None:
        out.node.mlp_first = h_pre.node
    return out


# ---- layer norm (+ residual, + stochastic depth) ---------------------------------------------------------------------
class _LayerNormNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        x, ln, sc = self.x, self.ln, self.shortcut
        rows, cp = _rows(x), x.cp
        c = x.c
        need_param = ln.weight.requires_grad or ln.bias.requires_grad
        if x.requires_grad or need_param:
            nrows = lib.tok_layernorm_bwd_rows(rows, c)
            partial = torch.empty((2, nrows, c), dtype=F32, device=g.device)
            if x.requires_grad:
                tgt, acc = grad_target(x)
            else:
                tgt, acc = torch.empty_like(x.data), 0
            _C.check(lib.tok_layernorm_bwd(ptr(g), ptr(x.data), ptr(self.mean), ptr(self.rstd), ptr(ln.weight),
                                           ptr(self.row_scale), self.rps, ptr(tgt), acc, ptr(partial), rows, c, cp, st),
                     'tok_layernorm_bwd')
            def run_param_grads():
                st_ = stream_ptr()
                keep = []
                if ln.weight.requires_grad and ln.bias.requires_grad:
                    (sw, mw), (sb, mb) = param_grad_target(ln.weight), param_grad_target(ln.bias)
                    if mw != 2 and mb != 2:        # both folds in one launch
                        _C.check(lib.tok_colsum_f32_pair(ptr(partial[0]), ptr(partial[1]), nrows, c, ptr(sw), 1 if mw == 1 else 0,
                                                         ptr(sb), 1 if mb == 1 else 0, st_), 'tok_colsum_f32_pair')
                        commit_param_grad(ln.weight, sw, mw)
                        commit_param_grad(ln.bias, sb, mb)
                        return keep
                for p, part in ((ln.weight, partial[0]), (ln.bias, partial[1])):
                    if p.requires_grad:
                        slot, mode = param_grad_target(p)
                        if mode == 2:
                            tmp = torch.empty_like(slot)
                            _C.check(lib.tok_colsum_f32(ptr(part), nrows, c, ptr(tmp), 0, st_), 'tok_colsum_f32')
                            p.grad.add_(tmp)
                            keep.append(tmp)
                            commit_param_grad(p, slot, 1)
                        else:
                            _C.check(lib.tok_colsum_f32(ptr(part), nrows, c, ptr(slot), 1 if mode == 1 else 0, st_),
                                     'tok_colsum_f32')
                            commit_param_grad(p, slot, mode)
                return keep
            if _use_side(self, g):
                with self.region.fork_side((partial,)):
                    self.region.keep_until_join(*run_param_grads())
            else:
                run_param_grads()
        if sc is not None and sc.requires_grad:
            # the residual branch passes the gradient through: hand the buffer over when we own it
            if not (self.out.grad_owned and donate_grad(sc, g.view(sc.data.shape))):
                tgt, acc = grad_target(sc)
                _C.check(lib.tok_act_bwd(2, ptr(g), ptr(g), ptr(tgt), acc, g.numel(), st), 'tok_act_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = self.shortcut = self.mean = self.rstd = self.row_scale = None


def layer_norm(region: Region, x: TTensor, ln: nn.LayerNorm, shortcut: Optional[TTensor] = None,
               row_scale: Optional[torch.Tensor] = None, rows_per_sample: int = 0) -> TTensor:
    """out = shortcut + row_scale[sample] * LayerNorm(x)."""
    if ln.weight is None or ln.bias is None or len(ln.normalized_shape) != 1 or ln.normalized_shape[0] != x.c:
        raise NotImplementedError('layer_norm: affine LayerNorm over the channel dimension only')
    lib, st = _C.lib(), stream_ptr()
    rows, cp = _rows(x), x.cp
    dev = x.data.device
    out_data = torch.empty_like(x.data)
    mean = torch.empty(rows, dtype=F32, device=dev)
    rstd = torch.empty(rows, dtype=F32, device=dev)
    _C.check(lib.tok_layernorm_fwd(ptr(x.data), ptr(shortcut.data) if shortcut is not None else None, ptr(row_scale),
                                   rows_per_sample, ptr(ln.weight), ptr(ln.bias), ptr(out_data), ptr(mean), ptr(rstd),
                                   rows, x.c, cp, float(ln.eps), st), 'tok_layernorm_fwd')
    req = region.grad_mode and (x.requires_grad or ln.weight.requires_grad or ln.bias.requires_grad or
                                (shortcut is not None and shortcut.requires_grad))
    out = TTensor(out_data, x.c, requires_grad=req)
    if req:
        node = _LayerNormNode()
        node.x, node.out, node.ln, node.shortcut = x, out, ln, shortcut
        node.mean, node.rstd, node.row_scale, node.rps = mean, rstd, row_scale, rows_per_sample
        out.node = node
        if x.requires_grad:
            x.uses += 1
        if shortcut is not None and shortcut.requires_grad:
            shortcut.uses += 1
        region.add(node)
    return out


# ---- activations ----------------------------------------------------------------------------------------------------------
RELU, GELU = 0, 1


class _ActNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        tgt, acc = grad_target(self.x)
        _C.check(_C.lib().tok_act_bwd(self.kind, ptr(g), ptr(self.x.data), ptr(tgt), acc, g.numel(), stream_ptr()),
                 'tok_act_bwd')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def activation(region: Region, x: TTensor, kind: int, precomputed: Optional[torch.Tensor] = None) -> TTensor:
    """act(x); `precomputed` = act(x) already produced by the epilogue of the GEMM that made x."""
    if precomputed is not None:
        y = precomputed
    else:
        y = torch.empty_like(x.data)
        _C.check(_C.lib().tok_act_fwd(kind, ptr(x.data), ptr(y), y.numel(), stream_ptr()), 'tok_act_fwd')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, x.c, requires_grad=req)
    if req:
        node = _ActNode()
        node.x, node.out, node.kind = x, out, kind
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- continuous relative position bias ------------------------------------------------------------------------------------
class _CpbBiasNode(Node):
    """Backward of the position-bias unit.  The attention unit that consumed the bias leaves its per-workgroup partial
    rows of d(logits) and d(logit_scale) here (`scratch`, `dscale`, `mark`); their fixed-order folds, d(logit_scale) and
    the gather back onto the cpb_mlp output run on THIS unit's stream — a branch stream in SwinV2 (swin.py), so the
    dependent chain of the main stream carries none of these small launches."""
    needs_backward = True

    def backward(self):
        if self.scratch is None:
            return
        lib, st = _C.lib(), stream_ptr()
        await_mark(self.mark)
        rows, heads, n = self.scratch.shape[0], self.heads, self.n
        dev = self.scratch.device
        ls = self.logit_scale
        if ls is not None and ls.requires_grad:
            slot, mode = param_grad_target(ls)
            if mode == 2:
                tmp = torch.empty_like(slot)
                _C.check(lib.tok_colsum_f32(ptr(self.dscale), rows, heads, ptr(tmp), 0, st), 'tok_colsum_f32')
                ls.grad.add_(tmp)
                commit_param_grad(ls, slot, 1)
            else:
                _C.check(lib.tok_colsum_f32(ptr(self.dscale), rows, heads, ptr(slot), 1 if mode == 1 else 0, st),
                         'tok_colsum_f32')
                commit_param_grad(ls, slot, mode)
        if not self.table.requires_grad:
            return
        dbias = torch.empty(heads * n * n, dtype=F32, device=dev)
        _C.check(lib.tok_colsum_f32(ptr(self.scratch), rows, heads * n * n, ptr(dbias), 0, st), 'tok_colsum_f32')
        trows, ld = self.table.shape
        tgt, acc = grad_target(self.table)
        if acc:
            raise RuntimeError('cpb_bias: the cpb_mlp output has a single consumer')
        _C.check(lib.tok_cpb_bias_bwd(ptr(dbias), 0, ptr(self.table.data), ld, ptr(self.index), heads, n, trows, ptr(tgt), st),
                 'tok_cpb_bias_bwd')

    def release(self):
        self.table = self.index = self.bias = self.scratch = self.dscale = self.mark = self.logit_scale = None


def cpb_bias(region: Region, table: TTensor, index: torch.Tensor, heads: int, n_tokens: int):
    """(bias fp32 [heads][N][N], node).  The attention unit hands its d(logits) / d(logit_scale) partial rows back through
    the node (see _CpbBiasNode)."""
    rows, ld = table.shape
    bias = torch.empty((heads, n_tokens, n_tokens), dtype=F32, device=table.data.device)
    _C.check(_C.lib().tok_cpb_bias_fwd(ptr(table.data), ld, ptr(index), heads, n_tokens, ptr(bias), stream_ptr()),
             'tok_cpb_bias_fwd')
    node = None
    if region.grad_mode and table.requires_grad:
        node = _CpbBiasNode()
        node.table, node.index, node.heads, node.n, node.bias = table, index, heads, n_tokens, bias
        node.scratch = node.dscale = node.mark = node.logit_scale = None
        table.uses += 1
        region.add(node)
    return bias, node


# ---- window attention ---------------------------------------------------------------------------------------------------------
class _WindowAttnNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        b, h, w, c, heads, ws, shift = self.geo
        n = ws * ws
        nw = (h // ws) * (w // ws)
        qkv = self.qkv
        dev = g.device
        tgt, acc = grad_target(qkv)
        if acc:
            raise RuntimeError('window_attention: qkv has a single consumer')
        if self.logit_scale is None:       # plain scaled-dot-product windows (DaViT): no bias / scale gradients
            _C.check(lib.tok_window_attn_bwd(ptr(qkv.data), ptr(g), b, h, w, c, heads, ws, shift, qkv.cp, None, None, None,
                                             ptr(self.lse), ptr(tgt), None, None, st), 'tok_window_attn_bwd')
            if qkv.cp != 3 * c:
                tgt[:, 3 * c:] = 0
            self.out.grad = None
            return
        rows = lib.tok_window_attn_bwd_rows(b, h, w, heads, ws)
        scratch = torch.empty((rows, heads * n * n), dtype=F32, device=dev)
        dscale = torch.empty((rows, heads), dtype=F32, device=dev)
        _C.check(lib.tok_window_attn_bwd(ptr(qkv.data), ptr(g), b, h, w, c, heads, ws, shift, qkv.cp,
                                         ptr(self.logit_scale), ptr(self.bias), ptr(self.mask), ptr(self.lse), ptr(tgt),
                                         ptr(scratch), ptr(dscale), st), 'tok_window_attn_bwd')
        if qkv.cp != 3 * c:
            tgt[:, 3 * c:] = 0
        ls = self.logit_scale
        bn = self.bias_node
        if bn is not None:
            # folds and parameter gradients happen in the position-bias unit's backward (its own stream)
            bn.scratch, bn.dscale, bn.logit_scale, bn.mark = scratch, dscale, ls, written_mark()
        elif ls.requires_grad:
            slot, mode = param_grad_target(ls)
            if mode == 2:
                tmp = torch.empty_like(slot)
                _C.check(lib.tok_colsum_f32(ptr(dscale), rows, heads, ptr(tmp), 0, st), 'tok_colsum_f32')
                ls.grad.add_(tmp)
                commit_param_grad(ls, slot, 1)
            else:
                _C.check(lib.tok_colsum_f32(ptr(dscale), rows, heads, ptr(slot), 1 if mode == 1 else 0, st),
                         'tok_colsum_f32')
                commit_param_grad(ls, slot, mode)
        self.out.grad = None

    def release(self):
        self.qkv = self.out = self.bias = self.mask = self.lse = self.bias_node = None


def window_attention(region: Region, qkv: TTensor, geo: Tuple[int, int, int, int, int, int, int],
                     logit_scale: Optional[nn.Parameter] = None, bias: Optional[torch.Tensor] = None, bias_node=None,
                     mask: Optional[torch.Tensor] = None) -> TTensor:
    """qkv rows [B*H*W][3C] -> attention output rows [B*H*W][C]; geo = (B, H, W, C, heads, window, shift).
    logit_scale / bias given: SwinV2 cosine attention; both None: softmax(q k^T / sqrt(head_dim)) v (DaViT)."""
    b, h, w, c, heads, ws, shift = geo
    if c != heads * 32:
        raise NotImplementedError('window_attention: head_dim 32 (every SwinV2 / DaViT variant of the reference)')
    if (logit_scale is None) != (bias is None):
        raise ValueError('window_attention: logit_scale and bias go together')
    lib, st = _C.lib(), stream_ptr()
    n = ws * ws
    nw = (h // ws) * (w // ws)
    dev = qkv.data.device
    out_data = torch.empty((b * h * w, c), dtype=BF16, device=dev)
    lse = torch.empty(b * nw * heads * n, dtype=F32, device=dev)
    _C.check(lib.tok_window_attn_fwd(ptr(qkv.data), b, h, w, c, heads, ws, shift, qkv.cp, ptr(logit_scale), ptr(bias),
                                     ptr(mask), ptr(out_data), ptr(lse), st), 'tok_window_attn_fwd')
    req = region.grad_mode and (qkv.requires_grad or (logit_scale is not None and logit_scale.requires_grad) or
                                bias_node is not None)
    out = TTensor(out_data, c, requires_grad=req)
    if req:
        node = _WindowAttnNode()
        node.qkv, node.out, node.geo, node.logit_scale = qkv, out, geo, logit_scale
        node.bias, node.bias_node, node.mask, node.lse = bias, bias_node, mask, lse
        out.node = node
        qkv.uses += 1
        region.add(node)
    return out


# ---- patch merging gather ----------------------------------------------------------------------------------------------------------
class _PatchMergeNode(Node):
    needs_backward = True

    def backward(self):
        g = self.out.grad
        if g is None or not self.x.requires_grad:
            return
        b, h, w, c = self.geo
        tgt, acc = grad_target(self.x)
        lib, st = _C.lib(), stream_ptr()
        if acc:      # the stage output also feeds its feature norm (forward_features): un-permute, then add
            tmp = torch.empty_like(tgt)
            _C.check(lib.tok_patch_merge(ptr(g), ptr(tmp), b, h, w, c, 1, st), 'tok_patch_merge')
            _C.check(lib.tok_act_bwd(2, ptr(tmp), ptr(tmp), ptr(tgt), 1, tmp.numel(), st), 'tok_act_bwd')
        else:
            _C.check(lib.tok_patch_merge(ptr(g), ptr(tgt), b, h, w, c, 1, st), 'tok_patch_merge')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def patch_merge(region: Region, x: TTensor, b: int, h: int, w: int) -> TTensor:
    """token rows [B*H*W][C] -> [B*(H/2)*(W/2)][4C] (x0, x1, x2, x3 concatenation of PatchMerging)."""
    c = x.c
    if c != x.cp:
        raise NotImplementedError('patch_merge: channels % 8 == 0')
    y = torch.empty((b * (h // 2) * (w // 2), 4 * c), dtype=BF16, device=x.data.device)
    _C.check(_C.lib().tok_patch_merge(ptr(x.data), ptr(y), b, h, w, c, 0, stream_ptr()), 'tok_patch_merge')
    req = region.grad_mode and x.requires_grad
    out = TTensor(y, 4 * c, requires_grad=req)
    if req:
        node = _PatchMergeNode()
        node.x, node.out, node.geo = x, out, (b, h, w, c)
        out.node = node
        x.uses += 1
        region.add(node)
    return out


# ---- DaViT: channel attention and the pre-norm residual ------------------------------------------------------------------------
class _ChannelAttnNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None or not self.qkv.requires_grad:
            return
        images, rpi, heads, c, scale = self.geo
        qkv, ld = self.qkv, self.qkv.cp
        q, k, v = (ptr(qkv.data) + 2 * o for o in (0, c, 2 * c))
        tgt, acc = grad_target(qkv)
        if acc:
            raise RuntimeError('channel_attention: qkv has a single consumer')
        dq, dk, dv = (ptr(tgt) + 2 * o for o in (0, c, 2 * c))
        ds = torch.empty_like(self.attn)
        # dA = dout^T q, through the softmax: dS = A o (dA - rowsum(dA o A))
        _C.check(lib.tok_chan_gram(ptr(g), g.stride(0), q, ld, rpi, images, heads, 1.0, 2, ptr(self.attn), ptr(ds), st),
                 'tok_chan_gram')
        _C.check(lib.tok_chan_apply(ptr(g), g.stride(0), ptr(self.attn), 1, 1.0, rpi, images, heads, dq, ld, st),
                 'tok_chan_apply')                                   # dq = dout A
        _C.check(lib.tok_chan_apply(v, ld, ptr(ds), 0, scale, rpi, images, heads, dk, ld, st), 'tok_chan_apply')   # dk
        _C.check(lib.tok_chan_apply(k, ld, ptr(ds), 1, scale, rpi, images, heads, dv, ld, st), 'tok_chan_apply')   # dv
        if ld != 3 * c:
            tgt[:, 3 * c:] = 0
        self.out.grad = None

    def release(self):
        self.qkv = self.out = self.attn = None


def channel_attention(region: Region, qkv: TTensor, images: int, rows_per_image: int, heads: int) -> TTensor:
    """davit.py:152-165 on qkv rows [B*N][3C]: A = softmax((k * scale)^T v) per (image, head), out = q A^T."""
    c = heads * 32
    if qkv.c != 3 * c:
        raise NotImplementedError('channel_attention: head_dim 32 (every DaViT variant of the reference)')
    lib, st = _C.lib(), stream_ptr()
    dev = qkv.data.device
    ld = qkv.cp
    scale = 32 ** -0.5
    q, k, v = (ptr(qkv.data) + 2 * o for o in (0, c, 2 * c))
    attn = torch.empty((images * heads, 32, 32), dtype=F32, device=dev)
    _C.check(lib.tok_chan_gram(k, ld, v, ld, rows_per_image, images, heads, scale, 1, None, ptr(attn), st), 'tok_chan_gram')
    out_data = torch.empty((images * rows_per_image, c), dtype=BF16, device=dev)
    _C.check(lib.tok_chan_apply(q, ld, ptr(attn), 0, 1.0, rows_per_image, images, heads, ptr(out_data), c, st),
             'tok_chan_apply')
    req = region.grad_mode and qkv.requires_grad
    out = TTensor(out_data, c, requires_grad=req)
    if req:
        node = _ChannelAttnNode()
        node.qkv, node.out, node.attn, node.geo = qkv, out, attn, (images, rows_per_image, heads, c, scale)
        out.node = node
        qkv.uses += 1
        region.add(node)
    return out


class _ResidualNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        a, b = self.a, self.b
        rows, cp = _rows(b), b.cp
        if b.requires_grad:
            tgt, acc = grad_target(b)
            _C.check(lib.tok_scale_rows_add(None, ptr(g), ptr(self.row_scale), self.rps, ptr(tgt), acc, rows, cp, st),
                     'tok_scale_rows_add')
        if a.requires_grad:
            if not (self.out.grad_owned and donate_grad(a, g.view(a.data.shape))):
                tgt, acc = grad_target(a)
                _C.check(lib.tok_act_bwd(2, ptr(g), ptr(g), ptr(tgt), acc, g.numel(), st), 'tok_act_bwd')
        self.out.grad = None

    def release(self):
        self.a = self.b = self.out = self.row_scale = None


def residual_add(region: Region, a: TTensor, b: TTensor, row_scale: Optional[torch.Tensor] = None,
                 rows_per_sample: int = 0) -> TTensor:
    """out = a + row_scale[sample] * b — `x + drop_path(f(x))` with the stochastic-depth keep/scale vector applied in place."""
    lib, st = _C.lib(), stream_ptr()
    rows, cp = _rows(b), b.cp
    if a.data.shape != b.data.shape:
        raise ValueError(f'residual_add: {tuple(a.data.shape)} vs {tuple(b.data.shape)}')
    out_data = torch.empty_like(b.data)
    _C.check(lib.tok_scale_rows_add(ptr(a.data), ptr(b.data), ptr(row_scale), rows_per_sample, ptr(out_data), 0, rows, cp, st),
             'tok_scale_rows_add')
    req = region.grad_mode and (a.requires_grad or b.requires_grad)
    out = TTensor(out_data, b.c, requires_grad=req)
    if req:
        node = _ResidualNode()
        node.a, node.b, node.out, node.row_scale, node.rps = a, b, out, row_scale, rows_per_sample
        out.node = node
        if a.requires_grad:
            a.uses += 1
        if b.requires_grad:
            b.uses += 1
        region.add(node)
    return out


# ---- DaViT ConvPosEnc with activation: depthwise 3x3 + bias ------------------------------------------------------------------
class _DwConvNode(Node):
    needs_backward = True

    def backward(self):
        lib, st = _C.lib(), stream_ptr()
        g = self.out.grad
        if g is None:
            return
        x, conv = self.x, self.conv
        n, h, w, ld = x.shape
        w_need, b_need = conv.weight.requires_grad, conv.bias is not None and conv.bias.requires_grad
        if w_need or b_need:
            blocks = lib.tok_dwconv3x3_wgrad_blocks(n, h)
            partial = torch.empty((blocks, x.c, 10), dtype=F32, device=g.device)
            ws, wm = param_grad_target(conv.weight) if w_need else (None, 0)
            bs, bm = param_grad_target(conv.bias) if b_need else (None, 0)
            if wm == 2 or bm == 2 or (w_need and b_need and wm != bm):
                raise NotImplementedError('dwconv3x3: foreign .grad tensors on the depthwise parameters')
            _C.check(lib.tok_dwconv3x3_wgrad(ptr(x.data), ptr(g), n, h, w, x.c, ld, ptr(partial), ptr(ws), ptr(bs),
                                             1 if (wm == 1 or bm == 1) else 0, st), 'tok_dwconv3x3_wgrad')
            if w_need:
                commit_param_grad(conv.weight, ws, wm)
            if b_need:
                commit_param_grad(conv.bias, bs, bm)
        if x.requires_grad:
            tgt, acc = grad_target(x)
            _C.check(lib.tok_dwconv3x3(ptr(g), ptr(conv.weight), None, ptr(tgt), acc, 1, n, h, w, x.c, ld, st), 'tok_dwconv3x3')
        self.out.grad = None

    def release(self):
        self.x = self.out = None


def dwconv3x3(region: Region, x: TTensor, conv: nn.Conv2d) -> TTensor:
    """Depthwise 3x3 / stride 1 / pad 1 convolution (+ bias) on a (B, H, W, C) map."""
    c = x.c
    if (conv.groups != c or conv.in_channels != c or conv.out_channels != c or tuple(conv.kernel_size) != (3, 3) or
            tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (1, 1) or not conv.weight.is_contiguous()):
        raise NotImplementedError('dwconv3x3: depthwise 3x3, stride 1, padding 1')
    n, h, w, ld = x.shape
    y = torch.empty_like(x.data)
    _C.check(_C.lib().tok_dwconv3x3(ptr(x.data), ptr(conv.weight), ptr(conv.bias), ptr(y), 0, 0, n, h, w, c, ld, stream_ptr()),
             'tok_dwconv3x3')
    req = region.grad_mode and (x.requires_grad or conv.weight.requires_grad)
    out = TTensor(y, c, requires_grad=req)
    if req:
        node = _DwConvNode()
        node.x, node.out, node.conv = x, out, conv
        out.node = node
        if x.requires_grad:
            x.uses += 1
        region.add(node)
    return out
