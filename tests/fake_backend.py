"""TEST-ONLY stand-in for libtok_gfx950.so operating on HOST memory with plain PyTorch ops.

It implements the C ABI of include/tok.h (same names, same argument order, same layouts and
rounding points) so that the host logic of torchok_amd — tape, gradient fan-in, parameter
arenas, optimizers, task wiring — can be exercised on a CPU-only box (`-m "not gpu"` tests).
Tests install it with `install()` at the bottom of this file — a monkeypatch applied from the
tests' side; the product has no hook for it and never routes through it (a missing native library raises).
"""
import ctypes
import math

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16
_SZ = {torch.float32: 4, torch.bfloat16: 2, torch.float16: 2, torch.int64: 8, torch.uint8: 1, torch.int32: 4}


def _t(ptr, shape, dtype):
    n = 1
    for s in shape:
        n *= int(s)
    if ptr is None or n == 0:
        return None
    buf = (ctypes.c_char * (n * _SZ[dtype])).from_address(int(ptr))
    return torch.frombuffer(buf, dtype=dtype, count=n).view(*[int(s) for s in shape])


def _desc(d):
    return d._obj if hasattr(d, '_obj') else d


def _bf(x):
    return x.to(BF16)


class FakeTok:
    """Each method == one extern "C" entry point."""

    def __init__(self):
        self.calls = []
        self._err = b''

    def tok_last_error(self):
        return self._err

    def tok_version(self):
        return 1

    # ---- layout ---------------------------------------------------------------------------------
    def tok_nchw_to_nhwc_bf16(self, src, dt, n, c, h, w, dst, c_pad, st):
        dtype = {0: torch.float32, 1: torch.float16, 2: BF16}[dt]
        s = _t(src, (n, c, h, w), dtype)
        d = _t(dst, (n, h, w, c_pad), BF16)
        d.zero_()
        d[..., :c] = s.permute(0, 2, 3, 1).to(BF16)
        return 0

    def tok_cast_f32_bf16(self, src, dst, count, st):
        _t(dst, (count,), BF16).copy_(_t(src, (count,), torch.float32))
        return 0

    def tok_cast_bf16_f32(self, src, dst, scale, count, st):
        _t(dst, (count,), torch.float32).copy_(_t(src, (count,), BF16).float() * scale)
        return 0

    def tok_pack_weight_fwd(self, src, k, r, s, c, dst, k_pad, s_pad, c_pad, st):
        w = _t(src, (k, r, s, c), torch.float32)
        d = _t(dst, (k_pad, r, s_pad, c_pad), BF16)
        d.zero_()
        d[:k, :, :s, :c] = w.to(BF16)
        return 0

    def tok_pack_weight_dgrad(self, src, k, r, s, c, dst, k_pad, c_pad, st):
        w = _t(src, (k, r, s, c), torch.float32)
        d = _t(dst, (c_pad, r, s, k_pad), BF16)
        d.zero_()
        d[:c, :, :, :k] = w.flip(1, 2).permute(3, 1, 2, 0).to(BF16)
        return 0

    def tok_pack_weight_both(self, src, k, r, s, c, dst_f, k_pad, s_pad, c_pad, dst_d, st):
        self.tok_pack_weight_fwd(src, k, r, s, c, dst_f, k_pad, s_pad, c_pad, st)
        return self.tok_pack_weight_dgrad(src, k, r, s, c, dst_d, k_pad, c_pad, st)

    def tok_pack_item_blocks(self, item):
        it = _desc(item)
        tf = it.k_pad * it.r * it.s_pad * it.c_pad if it.dst_fwd else 0
        td = it.c_pad * it.r * it.s * it.k_pad if it.dst_dgrad else 0
        return (max(tf, td) + 1023) // 1024

    def tok_pack_weights_batched(self, items, n_items, total_blocks, st):
        from torchok_amd._C import PackItem
        self.calls.append('pack_weights_batched')
        arr = ctypes.cast(ctypes.c_void_p(int(items)), ctypes.POINTER(PackItem))
        blocks = 0
        for i in range(n_items):
            it = arr[i]
            assert it.block_start == blocks
            blocks += self.tok_pack_item_blocks(it)
            if it.dst_fwd:
                self.tok_pack_weight_fwd(it.src, it.k, it.r, it.s, it.c, it.dst_fwd, it.k_pad, it.s_pad, it.c_pad, st)
            if it.dst_dgrad:
                self.tok_pack_weight_dgrad(it.src, it.k, it.r, it.s, it.c, it.dst_dgrad, it.k_pad, it.c_pad, st)
        assert blocks == total_blocks
        return 0

    # ---- conv -----------------------------------------------------------------------------------
    def tok_conv_fwd_stat_rows(self, d):
        d = _desc(d)
        return (d.n * d.p * d.q + 127) // 128

    def _weights(self, d, w):
        wt = _t(w, (d.k, d.r, d.s_pad, d.c), BF16).float()[:, :, :d.s, :]
        return wt.permute(0, 3, 1, 2).contiguous()  # K C R S

    def tok_conv_fwd(self, d, x, w, bias, y, stats, st):
        d = _desc(d)
        self.calls.append('conv_fwd')
        xin = _t(x, (d.n, d.h, d.w, d.c), BF16).float().permute(0, 3, 1, 2)
        b = _t(bias, (d.k,), torch.float32)
        out = F.conv2d(xin, self._weights(d, w), b, stride=d.stride, padding=d.pad)
        out = out.permute(0, 2, 3, 1).to(BF16)
        _t(y, (d.n, d.p, d.q, d.k), BF16).copy_(out)
        if stats is not None:
            rows = self.tok_conv_fwd_stat_rows(d)
            s = _t(stats, (2, rows, d.k), torch.float32)
            s.zero_()
            f = out.float().reshape(-1, d.k)
            s[0, 0] = f.sum(0)
            s[1, 0] = (f * f).sum(0)
        return 0

    def tok_conv_dgrad(self, d, dy, wd, dx, accumulate, st):
        d = _desc(d)
        self.calls.append('conv_dgrad')
        g = _t(dy, (d.n, d.p, d.q, d.k), BF16).float().permute(0, 3, 1, 2)
        pack = _t(wd, (d.c, d.r, d.s, d.k), BF16).float()
        wt = pack.flip(1, 2).permute(3, 0, 1, 2).contiguous()  # K C R S
        gi = torch.nn.grad.conv2d_input((d.n, d.c, d.h, d.w), wt, g.contiguous(), stride=d.stride, padding=d.pad)
        gi = gi.permute(0, 2, 3, 1)
        out = _t(dx, (d.n, d.h, d.w, d.c), BF16)
        if accumulate:
            out.copy_((gi + out.float()).to(BF16))
        else:
            out.copy_(gi.to(BF16))
        return 0

    def tok_conv_fwd_act(self, d, x, w, bias, y, y_act, kind, st):
        rc = self.tok_conv_fwd(d, x, w, bias, y, None, st)
        dd = _desc(d)
        return rc or self.tok_act_fwd(kind, y, y_act, dd.n * dd.p * dd.q * dd.k, st)

    def tok_mlp_serves(self, rows, c, hidden):
        import os
        return int(c in (96, 192, 384) and hidden == 4 * c and rows >= int(os.environ.get('TOK_MLP_MIN_ROWS', '32768')))

    # The fused Mlp entry points go through the SAME torch primitives, on the same shapes, as the launches they replace
    # (F.conv2d / conv2d_input on (rows, C, 1, 1) tensors, tok_act_fwd / tok_act_bwd's formulas): the host test compares the two
    # tapes bit for bit, and a matmul restatement rounds differently from oneDNN's convolution in the last bf16 place
    # depending on the thread count an earlier test module left behind.
    def tok_mlp_fwd(self, x, w1, b1, w2, b2, y, pre, act, rows, c, hidden, st):
        self.calls.append('mlp_fwd')
        xin = _t(x, (rows, 1, 1, c), BF16).float().permute(0, 3, 1, 2)
        p = F.conv2d(xin, _t(w1, (hidden, 1, 1, c), BF16).float().permute(0, 3, 1, 2).contiguous(), _t(b1, (hidden,), torch.float32))
        p = p.permute(0, 2, 3, 1).to(BF16).reshape(rows, hidden)
        pv = p.float()
        h = _bf(F.gelu(pv))
        hin = h.reshape(rows, 1, 1, hidden).float().permute(0, 3, 1, 2)
        o = F.conv2d(hin, _t(w2, (c, 1, 1, hidden), BF16).float().permute(0, 3, 1, 2).contiguous(), _t(b2, (c,), torch.float32))
        _t(y, (rows, c), BF16).copy_(o.permute(0, 2, 3, 1).to(BF16).reshape(rows, c))
        if pre is not None:
            _t(pre, (rows, hidden), BF16).copy_(p)
        if act is not None:
            _t(act, (rows, hidden), BF16).copy_(h)
        return 0

    def tok_mlp_bwd_dx(self, dy, w2d, pre, w1d, dx, accumulate, dpre, rows, c, hidden, st):
        self.calls.append('mlp_bwd_dx')

        def dgrad(g2d, pack2d, cin, k):             # tok_conv_dgrad on a (rows, 1, 1) map: pack [cin][1][1][k]
            g = g2d.reshape(rows, 1, 1, k).float().permute(0, 3, 1, 2)
            wt = pack2d.reshape(cin, 1, 1, k).float().flip(1, 2).permute(3, 0, 1, 2).contiguous()
            gi = torch.nn.grad.conv2d_input((rows, cin, 1, 1), wt, g.contiguous(), stride=1, padding=0)
            return gi.permute(0, 2, 3, 1).reshape(rows, cin)
        o = dgrad(_t(dy, (rows, c), BF16), _t(w2d, (hidden, c), BF16), hidden, c).to(BF16)
        v = _t(pre, (rows, hidden), BF16).float()
        d = 0.5 * (1 + torch.erf(v * 0.7071067811865476)) + v * 0.3989422804014327 * torch.exp(-0.5 * v * v)
        dp = _bf(o.float() * d)
        out = _t(dx, (rows, c), BF16)
        gi = dgrad(dp, _t(w1d, (c, hidden), BF16), c, hidden)
        out.copy_((gi + out.float()).to(BF16) if accumulate else gi.to(BF16))
        if dpre is not None:
            _t(dpre, (rows, hidden), BF16).copy_(dp)
        return 0

    def tok_conv_dgrad_act(self, d, dy, wd, act_x, kind, dx, st):
        rc = self.tok_conv_dgrad(d, dy, wd, dx, 0, st)
        dd = _desc(d)
        return rc or self.tok_act_bwd(kind, dx, act_x, dx, 0, dd.n * dd.h * dd.w * dd.c, st)

    def tok_conv_dgrad_stat_rows(self, d):
        return 2

    def tok_conv_dgrad_bnstats(self, d, dy, wd, dx, accumulate, bn_y, bn_mask, partial, st):
        rc = self.tok_conv_dgrad(d, dy, wd, dx, accumulate, st)
        d = _desc(d)
        self.calls.append('dgrad_bnstats')
        m = d.n * d.h * d.w
        g = _t(dx, (m, d.c), BF16).float()
        yv = _t(bn_y, (m, d.c), BF16).float()
        dz = g * self._bits(bn_mask, m, d.c) if bn_mask is not None else g
        p = _t(partial, (2, 2, d.c), torch.float32)
        p.zero_()
        p[0, 1] = dz.sum(0)
        p[1, 1] = (dz * yv).sum(0)
        return rc

    def tok_conv_fwd_bn(self, d, x, w, y, stats, bn, st):
        b = _desc(bn)
        rc = self.tok_conv_fwd(d, x, w, None, y, stats, st)
        dd = _desc(d)
        assert _t(b.counters, (64,), torch.int32).abs().sum() == 0
        return rc or self.tok_bn_finalize(stats, self.tok_conv_fwd_stat_rows(d), b.count, dd.k, b.c_real, b.gamma, b.beta, b.running_mean,
                                          b.running_var, b.nbt, b.momentum, b.eps, b.mean, b.rstd, b.scale, b.shift, st)

    def tok_conv_dgrad_bn(self, d, dy, wd, dx, accumulate, bn_y, bn_mask, partial, bn, st):
        b = _desc(bn)
        rc = self.tok_conv_dgrad_bnstats(d, dy, wd, dx, accumulate, bn_y, bn_mask, partial, st)
        dd = _desc(d)
        return rc or self.tok_bn_bwd_finalize(partial, self.tok_conv_dgrad_stat_rows(d), b.count, dd.c, b.c_real, b.gamma, b.mean, b.rstd, b.dgamma,
                                              b.dbeta, b.coef, b.param_accumulate, 1, st)

    # ---- "unit 3" (1x1 conv -> BatchNorm -> + shortcut -> ReLU without the pre-normalisation tensor) -----------------------
    def tok_bn_gram_finalize(self, Z, zsum, w, count, p, k, gamma, beta, rm, rv, nbt, momentum, eps, mean, rstd, scale, shift,
                             wz, st):
        self.calls.append('bn_gram_finalize')
        Zm = _t(Z, (p, p), torch.float32).double()
        mu_z = _t(zsum, (p,), torch.float32).double() / count
        wb = _t(w, (k, p), torch.float32).to(BF16).double()
        cov = Zm / count - torch.outer(mu_z, mu_z)
        _t(wz, (k, p), torch.float32).copy_((wb @ Zm).float())
        mu = wb @ mu_z
        var = ((wb @ cov) * wb).sum(1).clamp_min(0)
        g, b = _t(gamma, (k,), torch.float32), _t(beta, (k,), torch.float32)
        muf, rs = mu.float(), (1.0 / torch.sqrt(var + eps)).float()
        sc = g * rs
        for dst, val in ((mean, muf), (rstd, rs), (scale, sc), (shift, b - muf * sc)):
            _t(dst, (k,), torch.float32).copy_(val)
        if rm is not None:
            unb = count / (count - 1) if count > 1 else 1.0
            _t(rm, (k,), torch.float32).mul_(1 - momentum).add_(momentum * muf)
            _t(rv, (k,), torch.float32).mul_(1 - momentum).add_(momentum * (var * unb).float())
        if nbt is not None:
            _t(nbt, (1,), torch.int64).add_(1)
        return 0

    def tok_conv_fwd_bn_apply(self, d, x, w, scale, shift, shortcut, relu, out, mask, st):
        d = _desc(d)
        self.calls.append('conv_fwd_bn_apply')
        m = d.n * d.p * d.q
        xin = _t(x, (m, d.c), BF16).float()
        wt = _t(w, (d.k, d.c), BF16).float()
        z = (xin @ wt.t()) * _t(scale, (d.k,), torch.float32) + _t(shift, (d.k,), torch.float32)
        if shortcut is not None:
            z = z + _t(shortcut, (m, d.k), BF16).float()
        o = (z.clamp_min(0) if relu else z).to(BF16)
        _t(out, (m, d.k), BF16).copy_(o)
        if mask is not None:
            bits = (o.float() > 0).long().reshape(m, d.k // 8, 8)
            _t(mask, (m, d.k // 8), torch.uint8).copy_((bits << torch.arange(8)).sum(-1).to(torch.uint8))
        return 0

    def tok_conv_dgrad_maskstore(self, d, dy, wd, dx, accumulate, mask, partial, st):
        rc = self.tok_conv_dgrad(d, dy, wd, dx, accumulate, st)
        d = _desc(d)
        self.calls.append('dgrad_maskstore')
        m = d.n * d.h * d.w
        g = _t(dx, (m, d.c), BF16)
        dz = (g.float() * self._bits(mask, m, d.c)).to(BF16)
        g.copy_(dz)
        p = _t(partial, (2, 2, d.c), torch.float32)
        p.zero_()
        p[0, 1] = dz.float().sum(0)
        return rc

    def tok_conv_dgrad2_ok(self, d1, d2):
        a, b = _desc(d1), _desc(d2)
        return 1 if (a.r == 1 and b.r == 1 and a.c == b.c and a.c % 64 == 0 and (a.n, a.h, a.w) == (b.n, b.h, b.w)) else 0

    def tok_conv_dgrad2(self, d1, dy1, w1, d2, dy2, w2, bias, dx, accumulate, bn_y, bn_mask, partial, st):
        rc = self.tok_conv_dgrad(d1, dy1, w1, dx, accumulate, st)
        self.calls[-1] = 'conv_dgrad2'
        return rc or self.tok_conv_dgrad_bias(d2, dy2, w2, bias, dx, 1, bn_y, bn_mask, partial, st)

    # ---- stride-2 projection shortcut as a pointwise layer ---------------------------------------------------------------
    def tok_subsample2_fwd(self, x, n, h, w, c, out, st):
        self.calls.append('subsample2_fwd')
        p, q = (h + 1) // 2, (w + 1) // 2
        _t(out, (n, p, q, c), BF16).copy_(_t(x, (n, h, w, c), BF16)[:, ::2, ::2])
        return 0

    def tok_subsample2_bwd(self, dsub, n, h, w, c, dx, accumulate, st):
        self.calls.append('subsample2_bwd')
        p, q = (h + 1) // 2, (w + 1) // 2
        g = _t(dx, (n, h, w, c), BF16)
        s = _t(dsub, (n, p, q, c), BF16)
        if accumulate:
            g[:, ::2, ::2] = (g[:, ::2, ::2].float() + s.float()).to(BF16)
        else:
            g.zero_()
            g[:, ::2, ::2] = s
        return 0

    def tok_conv_dgrad_subacc_ok(self, d):
        d = _desc(d)
        return 1 if (d.r == 1 and d.s == 1 and d.stride == 1 and d.pad == 0 and d.c % 64 == 0) else 0

    def tok_conv_dgrad_subacc(self, d, dy, wd, dx, dsub, bn_y, mask, partial, mask_store, st):
        dd = _desc(d)
        self.calls.append('dgrad_subacc')
        p, q = (dd.h + 1) // 2, (dd.w + 1) // 2
        g = _t(dx, (dd.n, dd.h, dd.w, dd.c), BF16)
        g.zero_()
        g[:, ::2, ::2] = _t(dsub, (dd.n, p, q, dd.c), BF16)
        if mask_store:
            return self.tok_conv_dgrad_maskstore(d, dy, wd, dx, 1, mask, partial, st)
        if bn_y is not None:
            return self.tok_conv_dgrad_bnstats(d, dy, wd, dx, 1, bn_y, mask, partial, st)
        return self.tok_conv_dgrad(d, dy, wd, dx, 1, st)

    def tok_relu_mask_reduce(self, dout, mask, m, c, dz, partial, st):
        self.calls.append('relu_mask_reduce')
        g = _t(dout, (m, c), BF16).float()
        g = (g * self._bits(mask, m, c) if mask is not None else g).to(BF16)
        _t(dz, (m, c), BF16).copy_(g)
        p = _t(partial, (2, 1, c), torch.float32)
        p.zero_()
        p[0, 0] = g.float().sum(0)
        return 0

    def tok_bn3_bwd_prepare_ws_floats(self, p, k):
        return k * (p + 1)

    def tok_bn3_bwd_prepare(self, G, w, wz, zsum, partial, rows, count, p, k, gamma, mean, rstd, dgamma, dbeta, pacc, coef, dw,
                            dw_acc, wa, wb, cvec, ws, st):
        self.calls.append('bn3_bwd_prepare')
        Gm = _t(G, (k, p), torch.float32).double()
        wq = _t(w, (k, p), torch.float32).to(BF16).double()
        WZ, zs = _t(wz, (k, p), torch.float32).double(), _t(zsum, (p,), torch.float32).double()
        s1 = _t(partial, (2, rows, k), torch.float32)[0].double().sum(0)
        s2 = (Gm * wq).sum(1)
        mu, rs = _t(mean, (k,), torch.float32).double(), _t(rstd, (k,), torch.float32).double()
        sx = rs * (s2 - mu * s1)
        for ptr_, val in ((dgamma, sx.float()), (dbeta, s1.float())):
            if ptr_ is not None:
                t = _t(ptr_, (k,), torch.float32)
                t.copy_(t + val if pacc else val)
        g = _t(gamma, (k,), torch.float32)
        m1, m2 = (s1 / count).float(), (sx / count).float()
        c1 = g * rs.float()
        c2 = -c1 * rs.float() * m2
        c3 = -c1 * m1 - c2 * mu.float()
        co = _t(coef, (3, k), torch.float32)
        co[0], co[1], co[2] = c1, c2, c3
        dwv = (c1.double()[:, None] * Gm + c2.double()[:, None] * WZ + torch.outer(c3.double(), zs)).float()
        t = _t(dw, (k, p), torch.float32)
        t.copy_(t + dwv if dw_acc else dwv)
        _t(wa, (p, k), BF16).copy_((c1.double()[:, None] * wq).t().to(BF16))
        _t(wb, (p, p), BF16).copy_((wq.t() @ (c2.double()[:, None] * wq)).to(BF16))
        _t(cvec, (p,), torch.float32).copy_((c3.double() @ wq).float())
        return 0

    def tok_conv_dgrad_bias(self, d, dy, wd, bias, dx, accumulate, bn_y, bn_mask, partial, st):
        dd = _desc(d)
        m = dd.n * dd.h * dd.w
        self.calls.append('conv_dgrad')
        g = _t(dy, (dd.n, dd.p, dd.q, dd.k), BF16).float().permute(0, 3, 1, 2)
        pack = _t(wd, (dd.c, dd.r, dd.s, dd.k), BF16).float()
        wt = pack.flip(1, 2).permute(3, 0, 1, 2).contiguous()
        gi = torch.nn.grad.conv2d_input((dd.n, dd.c, dd.h, dd.w), wt, g.contiguous(), stride=dd.stride, padding=dd.pad)
        gi = gi.permute(0, 2, 3, 1)
        if bias is not None:
            gi = gi + _t(bias, (dd.c,), torch.float32)
        out = _t(dx, (dd.n, dd.h, dd.w, dd.c), BF16)
        out.copy_((gi + out.float()).to(BF16) if accumulate else gi.to(BF16))
        if bn_y is not None:
            self.calls.append('dgrad_bnstats')
            gv = _t(dx, (m, dd.c), BF16).float()
            yv = _t(bn_y, (m, dd.c), BF16).float()
            dz = gv * self._bits(bn_mask, m, dd.c) if bn_mask is not None else gv
            pt = _t(partial, (2, 2, dd.c), torch.float32)
            pt.zero_()
            pt[0, 1] = dz.sum(0)
            pt[1, 1] = (dz * yv).sum(0)
        return 0

    @staticmethod
    def _bits(mask, m, c):
        b = _t(mask, (m, c // 8), torch.uint8).long()
        return ((b.unsqueeze(-1) >> torch.arange(8)) & 1).reshape(m, c).float()

    def tok_conv_wgrad_ws_bytes(self, d):
        return 64

    def tok_conv_wgrad(self, d, x, dy, dw, k_real, c_real, ws, ws_bytes, accumulate, st):
        d = _desc(d)
        self.calls.append('conv_wgrad')
        xin = _t(x, (d.n, d.h, d.w, d.c), BF16).float().permute(0, 3, 1, 2).contiguous()
        g = _t(dy, (d.n, d.p, d.q, d.k), BF16).float().permute(0, 3, 1, 2).contiguous()
        gw = torch.nn.grad.conv2d_weight(xin, (d.k, d.c, d.r, d.s), g, stride=d.stride, padding=d.pad)
        gw = gw[:k_real, :c_real].permute(0, 2, 3, 1)  # k r s c
        out = _t(dw, (k_real, d.r, d.s, c_real), torch.float32)
        if accumulate:
            out.add_(gw)
        else:
            out.copy_(gw)
        return 0

    def tok_conv_wgrad_bias_ok(self, d):
        d = _desc(d)
        return 1 if (d.r == 1 and d.s == 1 and d.stride == 1 and d.pad == 0 and d.c != 4) else 0

    def tok_conv_wgrad_bias_ws_bytes(self, d):
        return self.tok_conv_wgrad_ws_bytes(d) + 4 * _desc(d).k

    def tok_conv_wgrad_bias(self, d, x, dy, dw, k_real, c_real, ws, ws_bytes, accumulate, dbias, bias_accumulate, st):
        rc = self.tok_conv_wgrad(d, x, dy, dw, k_real, c_real, ws, ws_bytes, accumulate, st)
        dd = _desc(d)
        self.calls.append('wgrad_bias')
        g = _t(dy, (dd.n * dd.p * dd.q, dd.k), BF16).float().sum(0)[:k_real]
        out = _t(dbias, (k_real,), torch.float32)
        if bias_accumulate:
            out.add_(g)
        else:
            out.copy_(g)
        return rc

    # ---- batch norm -------------------------------------------------------------------------------
    def tok_bn_finalize(self, stats, rows, count, cp, c, gamma, beta, rm, rv, nbt, momentum, eps, mean, rstd,
                        scale, shift, st):
        s = _t(stats, (2, rows, cp), torch.float32).double().sum(1)[:, :c]
        mu = s[0] / count
        var = (s[1] / count - mu * mu).clamp_min(0)
        g, b = _t(gamma, (c,), torch.float32), _t(beta, (c,), torch.float32)
        muf = mu.float()
        rs = (1.0 / torch.sqrt(var + eps)).float()
        sc = g * rs
        for dst, val in ((mean, muf), (rstd, rs), (scale, sc), (shift, b - muf * sc)):
            d = _t(dst, (cp,), torch.float32)
            d.zero_()
            d[:c] = val
        if rm is not None:
            unb = count / (count - 1) if count > 1 else 1.0
            m_, v_ = _t(rm, (c,), torch.float32), _t(rv, (c,), torch.float32)
            m_.mul_(1 - momentum).add_(momentum * muf)
            v_.mul_(1 - momentum).add_(momentum * (var * unb).float())
        if nbt is not None:
            _t(nbt, (1,), torch.int64).add_(1)
        return 0

    def tok_bn_eval_coeffs(self, gamma, beta, rm, rv, eps, cp, c, scale, shift, st):
        g, b = _t(gamma, (c,), torch.float32), _t(beta, (c,), torch.float32)
        m_, v_ = _t(rm, (c,), torch.float32), _t(rv, (c,), torch.float32)
        sc = g / torch.sqrt(v_ + eps)
        for dst, val in ((scale, sc), (shift, b - m_ * sc)):
            d = _t(dst, (cp,), torch.float32)
            d.zero_()
            d[:c] = val
        return 0

    def tok_bn_stats_rows(self, m, c):
        return 1

    def tok_bn_stats(self, y, m, c, stats, st):
        f = _t(y, (m, c), BF16).float()
        s = _t(stats, (2, 1, c), torch.float32)
        s[0, 0] = f.sum(0)
        s[1, 0] = (f * f).sum(0)
        return 0

    def tok_bn_act_fwd(self, y, scale, shift, shortcut, relu, out, mask, m, c, st):
        self.calls.append('bn_act_fwd')
        z = _t(y, (m, c), BF16).float() * _t(scale, (c,), torch.float32) + _t(shift, (c,), torch.float32)
        if shortcut is not None:
            z = z + _t(shortcut, (m, c), BF16).float()
        if relu:
            z = z.clamp_min(0)
        o = z.to(BF16)
        _t(out, (m, c), BF16).copy_(o)
        if mask is not None:
            bits = (o.float() > 0).long().reshape(m, c // 8, 8)
            _t(mask, (m, c // 8), torch.uint8).copy_((bits << torch.arange(8)).sum(-1).to(torch.uint8))
        return 0

    def tok_event_create(self):
        return 1

    def tok_event_destroy(self, ev):
        return 0

    def tok_next_launch_event(self, ev):
        return 0

    def tok_stream_wait_event(self, stream, ev):
        return 0

    def tok_bn_act_fwd_colsum_rows(self, m, c):
        return 3

    def tok_bn_act_fwd_colsum(self, y, scale, shift, shortcut, relu, out, mask, m, c, partial, st):
        rc = self.tok_bn_act_fwd(y, scale, shift, shortcut, relu, out, mask, m, c, st)
        self.calls[-1] = 'bn_act_fwd_colsum'
        p = _t(partial, (3, c), torch.float32)
        p.zero_()
        p[1] = _t(out, (m, c), BF16).float().sum(0)
        return rc

    def tok_bn_bwd_rows(self, m, c):
        return 1

    def _dz(self, dout, y, out, scale, shift, relu, m, c):
        g = _t(dout, (m, c), BF16).float()
        if not relu:
            return g
        if out is not None:     # `out` = the uint8 bit mask written by tok_bn_act_fwd
            mask = self._bits(out, m, c) > 0
        else:
            mask = (_t(y, (m, c), BF16).float() * _t(scale, (c,), torch.float32) + _t(shift, (c,), torch.float32)) > 0
        return g * mask

    def tok_bn_bwd_reduce(self, dout, y, out, scale, shift, mean, rstd, relu, m, c, partial, st):
        dz = self._dz(dout, y, out, scale, shift, relu, m, c)
        xhat = (_t(y, (m, c), BF16).float() - _t(mean, (c,), torch.float32)) * _t(rstd, (c,), torch.float32)
        p = _t(partial, (2, 1, c), torch.float32)
        p[0, 0] = dz.sum(0)
        p[1, 0] = (dz * xhat).sum(0)
        return 0

    def tok_bn_bwd_finalize(self, partial, rows, m, cp, c, gamma, mean, rstd, dgamma, dbeta, coef, accumulate, dzy,
                            st):
        p = _t(partial, (2, rows, cp), torch.float32).double().sum(1)[:, :c].clone()
        if dzy:
            p[1] = _t(rstd, (cp,), torch.float32)[:c].double() * (p[1] - _t(mean, (cp,), torch.float32)[:c].double() * p[0])
        sdz, sdzx = p[0].float(), p[1].float()
        for ptr_, val in ((dgamma, sdzx), (dbeta, sdz)):
            if ptr_ is not None:
                t = _t(ptr_, (c,), torch.float32)
                if accumulate:
                    t.add_(val)
                else:
                    t.copy_(val)
        g = _t(gamma, (c,), torch.float32)
        mu, rs = _t(mean, (cp,), torch.float32)[:c], _t(rstd, (cp,), torch.float32)[:c]
        m1, m2 = (p[0] / m).float(), (p[1] / m).float()
        c1 = g * rs
        c2 = -c1 * rs * m2
        co = _t(coef, (3, cp), torch.float32)
        co.zero_()
        co[0, :c], co[1, :c], co[2, :c] = c1, c2, -c1 * m1 - c2 * mu
        return 0

    def tok_bn_bwd_apply(self, dout, y, out, scale, shift, coef, relu, dy, dshortcut, ds_acc, m, c, st):
        self.calls.append('bn_bwd_apply')
        dz = self._dz(dout, y, out, scale, shift, relu, m, c)
        co = _t(coef, (3, c), torch.float32)
        res = co[0] * dz + co[1] * _t(y, (m, c), BF16).float() + co[2]
        _t(dy, (m, c), BF16).copy_(res.to(BF16))
        if dshortcut is not None:
            d = _t(dshortcut, (m, c), BF16)
            d.copy_(((dz + d.float()) if ds_acc else dz).to(BF16))
        return 0

    # ---- pooling ----------------------------------------------------------------------------------
    def tok_maxpool3x3s2_fwd(self, x, y, argmax, n, h, w, c, st):
        xin = _t(x, (n, h, w, c), BF16).float().permute(0, 3, 1, 2)
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out, idx = F.max_pool2d(xin, 3, 2, 1, return_indices=True)
        _t(y, (n, p, q, c), BF16).copy_(out.permute(0, 2, 3, 1).to(BF16))
        # flat input index -> tap index r*3+s
        ih, iw = idx // w, idx % w
        pp = torch.arange(p).view(1, 1, p, 1)
        qq = torch.arange(q).view(1, 1, 1, q)
        tap = (ih - (2 * pp - 1)) * 3 + (iw - (2 * qq - 1))
        _t(argmax, (n, p, q, c), torch.uint8).copy_(tap.permute(0, 2, 3, 1).to(torch.uint8))
        return 0

    def tok_maxpool3x3s2_bwd(self, dy, argmax, dx, accumulate, n, h, w, c, st):
        p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        g = _t(dy, (n, p, q, c), BF16).float()
        tap = _t(argmax, (n, p, q, c), torch.uint8).long()
        pp = torch.arange(p).view(1, p, 1, 1)
        qq = torch.arange(q).view(1, 1, q, 1)
        ih = 2 * pp - 1 + tap // 3
        iw = 2 * qq - 1 + tap % 3
        flat = (ih * w + iw)
        res = torch.zeros(n, h * w, c)
        res.scatter_add_(1, flat.view(n, p * q, c), g.view(n, p * q, c))
        out = _t(dx, (n, h, w, c), BF16)
        res = res.view(n, h, w, c)
        out.copy_(((res + out.float()) if accumulate else res).to(BF16))
        return 0

    # ---- stem: BatchNorm + ReLU + max-pool fused == the unfused entry points chained through scratch tensors ----------
    def tok_bn_relu_maxpool_fwd(self, y, scale, shift, n, h, w, c, pooled, argmax, ypool, st):
        z = torch.empty(n * h * w, c, dtype=BF16)
        self.tok_bn_act_fwd(y, scale, shift, None, 1, z.data_ptr(), None, n * h * w, c, st)
        rc = self.tok_maxpool3x3s2_fwd(z.data_ptr(), pooled, argmax, n, h, w, c, st)
        if ypool is not None:
            p, q = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            tap = _t(argmax, (n, p, q, c), torch.uint8).long()
            ih = 2 * torch.arange(p).view(1, p, 1, 1) - 1 + tap // 3
            iw = 2 * torch.arange(q).view(1, 1, q, 1) - 1 + tap % 3
            yv = _t(y, (n, h * w, c), BF16)
            _t(ypool, (n, p * q, c), BF16).copy_(torch.gather(yv, 1, (ih * w + iw).view(n, p * q, c)))
        return rc

    def tok_bn_pool_bwd_reduce_pooled(self, dpool, pooled, ypool, mean, rstd, mp, c, partial, st):
        g = _t(dpool, (mp, c), BF16).float() * (_t(pooled, (mp, c), BF16).float() > 0)
        xhat = (_t(ypool, (mp, c), BF16).float() - _t(mean, (c,), torch.float32)) * _t(rstd, (c,), torch.float32)
        p = _t(partial, (2, 1, c), torch.float32)
        p[0, 0] = g.sum(0)
        p[1, 0] = (g * xhat).sum(0)
        return 0

    def _pool_dz(self, dpool, argmax, y, scale, shift, n, h, w, c):
        m = n * h * w
        dz = torch.empty(m, c, dtype=BF16)
        self.tok_maxpool3x3s2_bwd(dpool, argmax, dz.data_ptr(), 0, n, h, w, c, None)
        z = torch.empty(m, c, dtype=BF16)
        mask = torch.empty(m, c // 8, dtype=torch.uint8)
        self.tok_bn_act_fwd(y, scale, shift, None, 1, z.data_ptr(), mask.data_ptr(), m, c, None)
        return dz, mask

    def tok_bn_pool_bwd_reduce(self, dpool, argmax, y, scale, shift, mean, rstd, n, h, w, c, partial, st):
        dz, mask = self._pool_dz(dpool, argmax, y, scale, shift, n, h, w, c)
        return self.tok_bn_bwd_reduce(dz.data_ptr(), y, mask.data_ptr(), scale, shift, mean, rstd, 1, n * h * w, c, partial, st)

    def tok_bn_pool_bwd_apply(self, dpool, argmax, y, scale, shift, coef, n, h, w, c, dy, st):
        dz, mask = self._pool_dz(dpool, argmax, y, scale, shift, n, h, w, c)
        return self.tok_bn_bwd_apply(dz.data_ptr(), y, mask.data_ptr(), scale, shift, coef, 1, dy, None, 0, n * h * w, c, st)

    def tok_avgpool2x2_fwd(self, x, y, n, h, w, c, st):
        xin = _t(x, (n, h, w, c), BF16).float().permute(0, 3, 1, 2)
        out = F.avg_pool2d(xin, 2, 2, ceil_mode=True, count_include_pad=False)
        _t(y, (n, (h + 1) // 2, (w + 1) // 2, c), BF16).copy_(_bf(out.permute(0, 2, 3, 1)))
        return 0

    def tok_avgpool2x2_bwd(self, dy, dx, accumulate, n, h, w, c, st):
        g = _t(dy, (n, (h + 1) // 2, (w + 1) // 2, c), BF16).float()
        hh, ww = torch.arange(h), torch.arange(w)
        cnt = (torch.where(2 * (hh // 2) + 1 < h, 2, 1)[:, None] * torch.where(2 * (ww // 2) + 1 < w, 2, 1)[None, :]).float()
        gx = g[:, hh // 2][:, :, ww // 2] / cnt[None, :, :, None]
        d = _t(dx, (n, h, w, c), BF16)
        d.copy_(_bf(d.float() + gx) if accumulate else _bf(gx))
        return 0

    def tok_gap_fwd(self, x, y, n, hw, c, st):
        _t(y, (n, c), BF16).copy_((_t(x, (n, hw, c), BF16).float().sum(1) * (1.0 / hw)).to(BF16))
        return 0

    def tok_gap_bwd(self, dy, dx, accumulate, n, hw, c, st):
        g = (_t(dy, (n, c), BF16).float() * (1.0 / hw)).unsqueeze(1).expand(n, hw, c)
        out = _t(dx, (n, hw, c), BF16)
        out.copy_(((g + out.float()) if accumulate else g).to(BF16))
        return 0

    def tok_global_pool_fwd(self, x, y, argmax, n, hw, c, ldy, mode, st):
        xv = _t(x, (n, hw, c), BF16).float()
        avg = (xv.sum(1) * (1.0 / hw)).to(BF16)
        mx, am = xv.max(1)
        # first maximal pixel (ATen keeps the earlier one on ties)
        first = (xv == mx[:, None, :]).float().argmax(1)
        _t(argmax, (n, c), torch.int32).copy_(first.to(torch.int32))
        out = _t(y, (n, ldy), BF16)
        if mode == 1:
            out[:, :c] = mx.to(BF16)
        elif mode == 2:
            out[:, :c] = (0.5 * (avg.float() + mx.to(BF16).float()).to(BF16).float()).to(BF16)
        else:
            out[:, :c] = avg
            out[:, c:2 * c] = mx.to(BF16)
        return 0

    def tok_global_pool_bwd(self, dy, argmax, dx, accumulate, n, hw, c, ldy, mode, st):
        g = _t(dy, (n, ldy), BF16).float()
        am = _t(argmax, (n, c), torch.int32).long()
        ga = g[:, :c]
        gm = g[:, c:2 * c] if mode == 3 else ga
        wa = 0.0 if mode == 1 else (0.5 / hw if mode == 2 else 1.0 / hw)
        wm = 0.5 if mode == 2 else 1.0
        grad = (ga * wa).unsqueeze(1).expand(n, hw, c).clone()
        grad.scatter_add_(1, am.unsqueeze(1), (gm * wm).unsqueeze(1))
        out = _t(dx, (n, hw, c), BF16)
        out.copy_(((grad + out.float()) if accumulate else grad).to(BF16))
        return 0

    def tok_colsum(self, dy, m, n_pad, n_real, out, accumulate, st):
        s = _t(dy, (m, n_pad), BF16).float().sum(0)[:n_real]
        o = _t(out, (n_real,), torch.float32)
        if accumulate:
            o.add_(s)
        else:
            o.copy_(s)
        return 0

    # ---- loss -------------------------------------------------------------------------------------
    def tok_softmax_ce_fwd(self, logits, target, rows, classes, ld, ignore_index, lse, row_loss, loss, st):
        return self.tok_softmax_ce_smooth_fwd(logits, target, rows, classes, ld, ignore_index, 0.0, lse, row_loss, loss, st)

    def tok_softmax_ce_smooth_fwd(self, logits, target, rows, classes, ld, ignore_index, smooth, lse, row_loss, loss, st):
        z = _t(logits, (rows, ld), BF16).float()[:, :classes]
        t = _t(target, (rows,), torch.int64)
        l = torch.logsumexp(z, 1)
        valid = (t != ignore_index) & (t >= 0) & (t < classes)
        tt = t.clamp(0, classes - 1)
        nll = l - z.gather(1, tt[:, None])[:, 0]
        if smooth:
            nll = (1 - smooth) * nll + smooth * (l - z.mean(1))
        rl = torch.where(valid, nll, torch.zeros(()))
        _t(lse, (rows,), torch.float32).copy_(l)
        _t(row_loss, (rows,), torch.float32).copy_(rl)
        o = _t(loss, (2,), torch.float32)
        nv = valid.sum().double()
        o[0] = (rl.double().sum() / nv).float()
        o[1] = nv.float()
        return 0

    # fused upsample + CE: the composition of the two restatements above (bf16-rounded interpolated logits, bf16-rounded
    # d(upsampled logits)), which is what the kernels compute without materialising either tensor
    def tok_upsample_ce_serves(self, classes, ld):
        return int(classes > 0 and ld >= classes and ld % 8 == 0 and ld <= 32)

    def _upsampled(self, low, n, hs, ws, classes, ld, hd, wd):
        x = _t(low, (n, hs, ws, ld), BF16)[..., :classes].float().permute(0, 3, 1, 2)
        y = F.interpolate(x, size=(hd, wd), mode='bilinear', align_corners=False)
        up = torch.zeros(n * hd * wd, ld, dtype=BF16)
        up[:, :classes] = _bf(y.permute(0, 2, 3, 1)).reshape(-1, classes)
        return up

    def tok_upsample_ce_fwd(self, low, n, hs, ws, classes, ld, hd, wd, target, ignore_index, lse, row_loss, loss, st):
        self.calls.append('upsample_ce_fwd')
        up = self._upsampled(low, n, hs, ws, classes, ld, hd, wd)
        return self.tok_softmax_ce_smooth_fwd(up.data_ptr(), target, n * hd * wd, classes, ld, ignore_index, 0.0, lse, row_loss,
                                              loss, st)

    def tok_upsample_ce_bwd(self, low, n, hs, ws, classes, ld, hd, wd, target, ignore_index, lse, loss, gscale, dlow,
                            accumulate, st):
        self.calls.append('upsample_ce_bwd')
        up = self._upsampled(low, n, hs, ws, classes, ld, hd, wd)
        dup = torch.zeros(n * hd * wd, ld, dtype=BF16)
        rc = self.tok_softmax_ce_smooth_bwd(up.data_ptr(), target, lse, loss, gscale, n * hd * wd, classes, ld, ignore_index, 0.0,
                                            dup.data_ptr(), st)
        calls = list(self.calls)
        rc = rc or self.tok_bilinear_bwd(dup.data_ptr(), n, hd, wd, ld, 0, dlow, hs, ws, classes, ld, accumulate, st)
        self.calls[:] = calls
        if not accumulate:
            _t(dlow, (n, hs, ws, ld), BF16)[..., classes:] = 0
        return rc

    def tok_softmax_ce_bwd(self, logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, dlogits, st):
        return self.tok_softmax_ce_smooth_bwd(logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, 0.0,
                                              dlogits, st)

    def tok_softmax_ce_smooth_bwd(self, logits, target, lse, loss, gscale, rows, classes, ld, ignore_index, smooth,
                                  dlogits, st):
        z = _t(logits, (rows, ld), BF16).float()[:, :classes]
        t = _t(target, (rows,), torch.int64)
        l = _t(lse, (rows,), torch.float32)
        nv = _t(loss, (2,), torch.float32)[1]
        gs = _t(gscale, (1,), torch.float32)[0] if gscale is not None else 1.0
        valid = (t != ignore_index) & (t >= 0) & (t < classes)
        p = torch.exp(z - l[:, None]) - smooth / classes
        p[torch.arange(rows)[valid], t[valid]] -= 1.0 - smooth
        p = p * (gs / nv) * valid[:, None]
        d = _t(dlogits, (rows, ld), BF16)        # d(logits) in the row pitch of the logits
        d.zero_()
        d[:, :classes] = p.to(BF16)
        return 0

    # ---- NT-Xent / triplet ------------------------------------------------------------------------------------------
    def _ntx(self, emb, n, d, ld, temperature):
        e = _t(emb, (n, ld), BF16)[:, :d].float()
        sim = (e @ e.t()) / temperature
        sim = sim.masked_fill(torch.eye(n, dtype=torch.bool), -1e9)
        labels = torch.cat([torch.arange(n // 2, n), torch.arange(n // 2)])
        return e, sim, labels

    def tok_ntxent_fwd(self, emb, n, d, ld, temperature, lse, row_loss, loss, st):
        _, sim, labels = self._ntx(emb, n, d, ld, temperature)
        l = torch.logsumexp(sim, 1)
        rl = l - sim[torch.arange(n), labels]
        _t(lse, (n,), torch.float32).copy_(l)
        _t(row_loss, (n,), torch.float32).copy_(rl)
        _t(loss, (1,), torch.float32)[0] = rl.mean()
        return 0

    def tok_ntxent_bwd(self, emb, lse, gscale, n, d, ld, temperature, demb, st):
        e, sim, labels = self._ntx(emb, n, d, ld, temperature)
        g = _t(gscale, (1,), torch.float32)[0] if gscale else 1.0
        w = torch.softmax(sim, 1) - F.one_hot(labels, n).float()
        w = w.masked_fill(torch.eye(n, dtype=torch.bool), 0.0)
        de = ((w + w.t()) @ e) * g / (n * temperature)
        o = _t(demb, (n, ld), BF16)
        o.zero_()
        o[:, :d] = _bf(de)
        return 0

    def _trip(self, a, p, ng, rows, d, ld, eps):
        av, pv, nv = (_t(q, (rows, ld), BF16)[:, :d].float() for q in (a, p, ng))
        return av, pv, nv, (av - pv + eps).norm(dim=1), (av - nv + eps).norm(dim=1), (pv - nv + eps).norm(dim=1)

    def tok_triplet_fwd(self, a, p, ng, rows, d, ld, margin, eps, swap, dist, row_loss, loss, st):
        _, _, _, dap, dan, dpn = self._trip(a, p, ng, rows, d, ld, eps)
        _t(dist, (rows, 3), torch.float32).copy_(torch.stack([dap, dan, dpn], 1))
        dneg = torch.minimum(dan, dpn) if swap else dan
        rl = (dap - dneg + margin).clamp_min(0)
        _t(row_loss, (rows,), torch.float32).copy_(rl)
        _t(loss, (1,), torch.float32)[0] = rl.mean()
        return 0

    def tok_triplet_bwd(self, a, p, ng, dist, gscale, rows, d, ld, margin, eps, swap, da, dp, dn, st):
        av, pv, nv, dap, dan, dpn = self._trip(a, p, ng, rows, d, ld, eps)
        g = (_t(gscale, (1,), torch.float32)[0] if gscale else 1.0) / rows
        sw = (dpn < dan) if swap else torch.zeros_like(dan, dtype=torch.bool)
        dneg = torch.where(sw, dpn, dan)
        act = ((dap - dneg + margin) > 0).float()[:, None] * g
        up = (av - pv + eps) / dap[:, None]
        un_a = (av - nv + eps) / dan[:, None]
        un_p = (pv - nv + eps) / dpn[:, None]
        swf = sw.float()[:, None]
        ga = up - (1 - swf) * un_a
        gp = -up - swf * un_p
        gn = (1 - swf) * un_a + swf * un_p
        for ptr_, val in ((da, ga), (dp, gp), (dn, gn)):
            o = _t(ptr_, (rows, ld), BF16)
            o.zero_()
            o[:, :d] = _bf(val * act)
        return 0

    # ---- object-contextual representations --------------------------------------------------------------------------------
    @staticmethod
    def _strided(p, rows, ld, c, dtype=BF16):
        flat = _t(p, ((rows - 1) * ld + c,), dtype)
        return torch.as_strided(flat, (rows, c), (ld, 1))

    def tok_pix_class_matmul(self, x, ldx, m, ldm, images, n, k, c, scale, out, st):
        xv = self._strided(x, images * n, ldx, c).float().view(images, n, c)
        mv = self._strided(m, images * k, ldm, c).float().view(images, k, c)
        _t(out, (images, n, k), torch.float32).copy_(scale * xv @ mv.transpose(1, 2))
        return 0

    def tok_class_pix_expand(self, w, m, ldm, images, n, k, c, scale, out, ldo, accumulate, st):
        wv = _t(w, (images, n, k), torch.float32)
        mv = self._strided(m, images * k, ldm, c).float().view(images, k, c)
        o = _t(out, (images * n, ldo), BF16)
        res = (scale * wv @ mv).reshape(images * n, c)
        cp = (c + 7) // 8 * 8
        if accumulate:
            o[:, :c] = _bf(res + o[:, :c].float())
        else:
            o[:, :cp] = 0
            o[:, :c] = _bf(res)
        return 0

    def tok_weighted_pool_chunks(self, n):
        return max(1, (n + 511) // 512)

    def tok_weighted_pool(self, w, x, ldx, images, n, k, c, scale, partial, out, ldo, accumulate, st):
        wv = _t(w, (images, n, k), torch.float32)
        xv = self._strided(x, images * n, ldx, c).float().view(images, n, c)
        res = (scale * wv.transpose(1, 2) @ xv).reshape(images * k, c)
        o = self._strided(out, images * k, ldo, c)
        o.copy_(_bf(res + (o.float() if accumulate else 0)))
        return 0

    def tok_softmax_rows_f32(self, x, rows, k, out, st):
        _t(out, (rows, k), torch.float32).copy_(_t(x, (rows, k), torch.float32).softmax(-1))
        return 0

    def tok_softmax_rows_bwd_f32(self, p, dp, rows, k, dx, st):
        pv, dv = _t(p, (rows, k), torch.float32), _t(dp, (rows, k), torch.float32)
        _t(dx, (rows, k), torch.float32).copy_(pv * (dv - (pv * dv).sum(-1, keepdim=True)))
        return 0

    def tok_softmax_cols_fwd(self, logits, ld, images, n, k, scale, p, st):
        lv = self._strided(logits, images * n, ld, k).float().view(images, n, k)
        _t(p, (images, n, k), torch.float32).copy_((scale * lv).softmax(1))
        return 0

    def tok_softmax_cols_bwd(self, p, dp, images, n, k, scale, dlogits, ld, accumulate, st):
        pv, dv = _t(p, (images, n, k), torch.float32), _t(dp, (images, n, k), torch.float32)
        res = scale * pv * (dv - (pv * dv).sum(1, keepdim=True))
        o = _t(dlogits, (images * n, ld), BF16)
        full = torch.zeros(images * n, ld)
        full[:, :k] = res.reshape(images * n, k)
        o.copy_(_bf(full + (o.float() if accumulate else 0)))
        return 0

    def tok_channel_scale(self, x, s, out, accumulate, images, n, c, ld, st):
        xv = _t(x, (images, n, ld), BF16).float()
        sv = torch.zeros(images, 1, ld)
        sv[:, 0, :c] = _t(s, (images, c), torch.float32)
        o = _t(out, (images, n, ld), BF16)
        o.copy_(_bf(xv * sv + (o.float() if accumulate else 0)))
        return 0

    # ---- depthwise 3x3 (DaViT ConvPosEnc with activation) -----------------------------------------------------------
    def tok_dwconv3x3(self, x, w, bias, out, accumulate, flip, n, h, wd, c, ld, st):
        xv = _t(x, (n, h, wd, ld), BF16)[..., :c].float().permute(0, 3, 1, 2)
        wv = _t(w, (c, 1, 3, 3), torch.float32)
        if flip:
            wv = wv.flip(2, 3)
        b = _t(bias, (c,), torch.float32) if bias is not None else None
        y = F.conv2d(xv, wv, b, padding=1, groups=c).permute(0, 2, 3, 1)
        o = _t(out, (n, h, wd, ld), BF16)
        if accumulate:
            o[..., :c] = _bf(y + o[..., :c].float())
        else:
            o.zero_()
            o[..., :c] = _bf(y)
        return 0

    def tok_dwconv3x3_wgrad_blocks(self, n, h):
        return 1

    def tok_dwconv3x3_wgrad(self, x, dout, n, h, wd, c, ld, partial, dw, db, accumulate, st):
        xv = _t(x, (n, h, wd, ld), BF16)[..., :c].float().permute(0, 3, 1, 2).contiguous()
        gv = _t(dout, (n, h, wd, ld), BF16)[..., :c].float().permute(0, 3, 1, 2).contiguous()
        gw = torch.nn.grad.conv2d_weight(xv, (c, 1, 3, 3), gv, padding=1, groups=c).reshape(c, 9)
        if dw is not None:
            t = _t(dw, (c, 9), torch.float32)
            t.copy_(gw + (t if accumulate else 0))
        if db is not None:
            t = _t(db, (c,), torch.float32)
            t.copy_(gv.sum((0, 2, 3)) + (t if accumulate else 0))
        return 0

    # ---- retrieval meters ------------------------------------------------------------------------------------------------
    def tok_sim_matrix(self, q, g, nq, ng, d, ldq, ldg, metric, out, ldo, st):
        qv = _t(q, (nq, ldq), torch.float32)[:, :d]
        gv = _t(g, (ng, ldg), torch.float32)[:, :d]
        s = qv @ gv.t() if metric == 0 else -((qv[:, None, :] - gv[None, :, :]) ** 2).sum(-1)
        _t(out, (nq, ldo), torch.float32)[:, :ng] = s
        return 0

    def tok_topk_rows(self, s, rows, cols, ld, k, vals, idx, st):
        sv = _t(s, (rows, ld), torch.float32)[:, :cols]
        order = torch.argsort(-sv, dim=1, stable=True)[:, :k]
        v = torch.gather(sv, 1, order)
        vo, io = _t(vals, (rows, k), torch.float32), _t(idx, (rows, k), torch.int64)
        vo.fill_(float('-inf'))
        io.fill_(-1)
        vo[:, :order.shape[1]] = v
        io[:, :order.shape[1]] = order
        return 0

    def tok_retrieval_nrel(self, labels, scores, n, n_cols, q_row, q_col, nq, n_rel, st):
        out = _t(n_rel, (nq,), torch.int32)
        if labels is not None:
            lab, rows = _t(labels, (n,), torch.int64), _t(q_row, (nq,), torch.int64)
            out.copy_(((lab[None, :] == lab[rows][:, None]).sum(1) - 1).int())
        else:
            sc, cols = _t(scores, (n, n_cols), torch.float32), _t(q_col, (nq,), torch.int64)
            out.copy_((sc[:, cols] >= 1).sum(0).int())
        return 0

    def tok_retrieval_eval(self, kind, idx, kk, drop_first, gallery, ng, labels, scores, n_cols, q_row, q_col, n_rel,
                           ideal, nq, out, st):
        import math
        iv = _t(idx, (nq, kk), torch.int64)
        df = _t(drop_first, (nq,), torch.uint8)
        gal = _t(gallery, (ng,), torch.int64) if gallery is not None else None
        rows = _t(q_row, (nq,), torch.int64)
        nr = _t(n_rel, (nq,), torch.int32)
        k = kk - 1
        n_all = int(max(int(rows.max()), int(gal.max()) if gal is not None else ng - 1)) + 1
        lab = _t(labels, (n_all,), torch.int64) if labels is not None else None
        cols = _t(q_col, (nq,), torch.int64) if labels is None else None
        idl = _t(ideal, (nq, k), torch.float32) if ideal is not None else None
        o = _t(out, (nq,), torch.float32)
        for qi in range(nq):
            if int(nr[qi]) == 0 or k <= 0:
                o[qi] = 0.
                continue
            first = 1 if int(df[qi]) else 0
            hits = ap = dcg = 0.0
            for p_ in range(k):
                li = int(iv[qi, first + p_])
                li = ng - 1 if li < 0 else li
                gi = int(gal[li]) if gal is not None else li
                if lab is not None:
                    gain = 1.0 if (int(lab[gi]) == int(lab[int(rows[qi])]) and gi != int(rows[qi])) else 0.0
                else:
                    # scores has n_all rows at least up to gi; address it flat
                    gain = float(_t(scores, (gi + 1, n_cols), torch.float32)[gi, int(cols[qi])])
                    gain = gain if gain >= 1 else 0.0
                if gain > 0:
                    hits += 1
                    ap += hits / (p_ + 1)
                    dcg += gain / math.log2(p_ + 2)
            n_r = int(nr[qi])
            if kind == 0:
                v = 1.0 if hits > 0 else 0.0
            elif kind == 1:
                v = hits / k
            elif kind == 2:
                v = hits / n_r
            elif kind == 3:
                v = ap / n_r
            else:
                ki = min(k, n_r)
                idcg = sum((float(idl[qi, p_]) if idl is not None else 1.0) / math.log2(p_ + 2) for p_ in range(ki))
                v = dcg / idcg
            o[qi] = v
        return 0

    def tok_cls_stats_update(self, logits, labels, target, rows, classes, ld, ignore_index, counts, st):
        t = _t(target, (rows,), torch.int64)
        pred = _t(labels, (rows,), torch.int64) if labels is not None else _t(logits, (rows, ld), BF16)[:, :classes].float().argmax(1)
        ok = (t != ignore_index) & (t >= 0) & (t < classes)
        cnt = _t(counts, (3, classes), torch.int64)
        for c in range(classes):
            cnt[0, c] += int(((pred == c) & (t == c) & ok).sum())
            cnt[1, c] += int(((pred == c) & ok).sum())
            cnt[2, c] += int(((t == c) & ok).sum())
        return 0

    def tok_confusion_update(self, logits, labels, target, rows, classes, ld, ignore_index, confusion, st):
        t = _t(target, (rows,), torch.int64)
        pred = _t(labels, (rows,), torch.int64) if labels is not None else _t(logits, (rows, ld), BF16)[:, :classes].float().argmax(1)
        ok = (t != ignore_index) & (t >= 0) & (t < classes) & (pred >= 0) & (pred < classes)
        conf = _t(confusion, (classes, classes), torch.int64)
        conf += torch.bincount(t[ok] * classes + pred[ok], minlength=classes * classes).view(classes, classes)
        return 0

    # ---- Dice loss ----------------------------------------------------------------------------------------------
    def tok_dice_rows(self, rows):
        return 1

    @staticmethod
    def _dice_probs(logits, target, rows, classes, ld, mode):
        z = _t(logits, (rows, ld), BF16)[:, :classes].float()
        if mode == 0:
            return z.softmax(1), F.one_hot(_t(target, (rows,), torch.int64), classes).float()
        if mode == 2:
            return torch.sigmoid(z), _t(target, (rows, classes), torch.float32)
        return torch.sigmoid(z), _t(target, (rows,), torch.float32)[:, None]

    def tok_dice_fwd(self, logits, target, rows, classes, ld, mode, smooth, eps, log_loss, sel, n_sel, partial, loss,
                     coef, st):
        p, y = self._dice_probs(logits, target, rows, classes, ld, mode)
        I, P, Y = (p * y).sum(0), p.sum(0), y.sum(0)
        pr = _t(partial, (1, 3, classes), torch.float32)
        pr[0, 0], pr[0, 1], pr[0, 2] = I, P, Y
        card = P + Y
        den = card.clamp_min(eps) + smooth
        num = 2 * I + smooth
        score = num / den
        counted = torch.ones(classes, dtype=torch.bool)
        ncount = classes
        if sel is not None:
            idx = _t(sel, (n_sel,), torch.int64)
            counted = torch.zeros(classes, dtype=torch.bool)
            counted[idx] = True
            ncount = n_sel
        act = counted & (Y > 0)
        if log_loss:
            l = -torch.log(score.clamp_min(eps))
            dl_ds = torch.where(score > eps, -1 / score, torch.zeros_like(score))
        else:
            l = 1 - score
            dl_ds = -torch.ones_like(score)
        _t(loss, (1,), torch.float32)[0] = (l * act).sum() / ncount
        co = _t(coef, (2, classes), torch.float32)
        co[0] = torch.where(act, dl_ds * 2 / den / ncount, torch.zeros_like(score))
        co[1] = torch.where(act & (card > eps), dl_ds * (-num / den ** 2) / ncount, torch.zeros_like(score))
        return 0

    def tok_dice_bwd(self, logits, target, coef, gscale, rows, classes, ld, mode, dlogits, st):
        p, y = self._dice_probs(logits, target, rows, classes, ld, mode)
        co = _t(coef, (2, classes), torch.float32)
        g = _t(gscale, (1,), torch.float32)[0] if gscale else 1.0
        dp = co[0] * y + co[1]
        dz = p * (dp - (p * dp).sum(1, keepdim=True)) if mode == 0 else p * (1 - p) * dp
        d = _t(dlogits, (rows, ld), BF16)
        d.zero_()
        d[:, :classes] = _bf(dz * g)
        return 0

    @staticmethod
    def _reg(kind, x, t, k):
        fn = {0: lambda: F.l1_loss(x, t, reduction='none'), 1: lambda: F.mse_loss(x, t, reduction='none'),
              2: lambda: F.smooth_l1_loss(x, t, reduction='none', beta=k),
              3: lambda: F.huber_loss(x, t, reduction='none', delta=k)}[kind]
        return fn()

    def tok_regression_loss_fwd(self, x, target, n, kind, knee, mean, loss, st):
        xv, t = _t(x, (n,), BF16).float(), _t(target, (n,), torch.float32)
        el = self._reg(kind, xv, t, knee).double()
        o = _t(loss, (2,), torch.float32)
        o[0] = float(el.mean() if mean else el.sum())
        o[1] = float(n)
        return 0

    def tok_regression_loss_bwd(self, x, target, gscale, n, kind, knee, mean, dx, st):
        t = _t(target, (n,), torch.float32)
        with torch.enable_grad():
            xv = _t(x, (n,), BF16).float().requires_grad_(True)
            self._reg(kind, xv, t, knee).sum().backward()
        g = float(_t(gscale, (1,), torch.float32)[0]) * (1.0 / n if mean else 1.0)
        _t(dx, (n,), BF16).copy_(_bf(xv.grad * g))
        return 0

    def tok_bce_logits_fwd(self, logits, target, rows, classes, ld, ignore_value, mean, loss, st):
        x = _t(logits, (rows, ld), BF16).float()[:, :classes]
        t = _t(target, (rows, classes), torch.float32)
        sel = t != ignore_value
        el = ((1 - t) * x - torch.nn.functional.logsigmoid(x))[sel].double()
        o = _t(loss, (2,), torch.float32)
        n = int(sel.sum())
        o[0] = float((el.sum() / n if mean else el.sum())) if n else 0.0
        o[1] = float(n)
        return 0

    def tok_bce_logits_bwd(self, logits, target, loss, gscale, rows, classes, ld, ignore_value, mean, dlogits, st):
        x = _t(logits, (rows, ld), BF16).float()[:, :classes]
        t = _t(target, (rows, classes), torch.float32)
        n = float(_t(loss, (2,), torch.float32)[1])
        g = float(_t(gscale, (1,), torch.float32)[0]) * ((1.0 / n if n else 0.0) if mean else 1.0)
        d = _t(dlogits, (rows, ld), BF16)
        d.zero_()
        d[:, :classes] = _bf(torch.where(t != ignore_value, (torch.sigmoid(x) - t) * g, torch.zeros(())))
        return 0

    # ---- metric-learning head / loss ---------------------------------------------------------------
    def tok_l2norm_fwd(self, x, y, inv_norm, rows, c, ld, is_f32, eps, st):
        dt = torch.float32 if is_f32 else BF16
        xs = _t(x, (rows, ld), dt)[:, :c].float()
        inv = 1.0 / xs.pow(2).sum(1).sqrt().clamp_min(eps)
        _t(inv_norm, (rows,), torch.float32).copy_(inv)
        out = _t(y, (rows, ld), dt)
        out.zero_()
        out[:, :c] = (xs * inv[:, None]).to(dt)
        return 0

    def tok_l2norm_bwd(self, dy, y, inv_norm, dx, accumulate, rows, c, ld, is_f32, st):
        dt = torch.float32 if is_f32 else BF16
        g = _t(dy, (rows, ld), dt)[:, :c].float()
        yy = _t(y, (rows, ld), dt)[:, :c].float()
        inv = _t(inv_norm, (rows,), torch.float32)
        v = (g - yy * (g * yy).sum(1, keepdim=True)) * inv[:, None]
        out = _t(dx, (rows, ld), dt)
        if accumulate:
            out[:, :c] = (out[:, :c].float() + v).to(dt)
        else:
            out.zero_()
            out[:, :c] = v.to(dt)
        return 0

    def tok_arcface_margin_fwd(self, cosine, target, rows, classes, ld, cos_m, sin_m, th, mm, easy, scale, out, st):
        cs = _t(cosine, (rows, ld), BF16)[:, :classes].float()
        t = _t(target, (rows,), torch.int64)
        sn = (1.0 - cs * cs).clamp(0, 1).sqrt()
        phi = _bf(cs * cos_m - sn * sin_m).float()
        phi = torch.where(cs > 0, phi, cs) if easy else torch.where(cs > th, phi, cs - mm)
        oh = F.one_hot(t, classes).bool()
        o = _t(out, (rows, ld), BF16)
        o.zero_()
        o[:, :classes] = _bf(torch.where(oh, phi, cs) * scale)
        return 0

    def tok_arcface_margin_bwd(self, cosine, target, dout, rows, classes, ld, cos_m, sin_m, th, mm, easy, scale,
                               dcos, st):
        cs = _t(cosine, (rows, ld), BF16)[:, :classes].float()
        t = _t(target, (rows,), torch.int64)
        g = _t(dout, (rows, ld), BF16)[:, :classes].float()
        q = 1.0 - cs * cs
        dsn = torch.where((q > 0) & (q < 1), -cs / q.clamp_min(1e-30).sqrt(), torch.zeros_like(cs))
        use_phi = (cs > 0) if easy else (cs > th)
        d = torch.where(use_phi, cos_m - dsn * sin_m, torch.ones_like(cs))
        d = torch.where(F.one_hot(t, classes).bool(), d, torch.ones_like(cs))
        o = _t(dcos, (rows, ld), BF16)
        o.zero_()
        o[:, :classes] = _bf(g * scale * d)
        return 0

    def tok_relevance_matrix(self, la, lb, na, nb, R, st):
        a, b = _t(la, (na,), torch.int64), _t(lb, (nb,), torch.int64)
        _t(R, (na, nb), torch.float32).copy_((a[:, None] == b[None, :]).float())
        return 0

    def tok_relevance_matrix_multilabel(self, ya, yb, na, nb, classes, R, st):
        a, b = _t(ya, (na, classes), torch.float32), _t(yb, (nb, classes), torch.float32)
        _t(R, (na, nb), torch.float32).copy_((a @ b.t() > 0).float())
        return 0

    def tok_embed_reg_fwd(self, e, n, d, ld, mode, row_reg, out, st):
        x = _t(e, (n, ld), BF16).float()[:, :d]
        rr = x.abs().sum(1) if mode == 1 else x.pow(2).sum(1).sqrt()
        _t(row_reg, (n,), torch.float32).copy_(rr)
        _t(out, (1,), torch.float32)[0] = float(rr.double().mean())
        return 0

    def tok_embed_reg_bwd(self, e, row_reg, gscale, coeff, n, d, ld, mode, de, st):
        x = _t(e, (n, ld), BF16).float()[:, :d]
        g = float(_t(gscale, (1,), torch.float32)[0]) * coeff
        rr = _t(row_reg, (n,), torch.float32)
        v = torch.sign(x) * g if mode == 1 else torch.where(rr[:, None] > 0, g * x / rr[:, None], torch.zeros(()))
        out = _t(de, (n, ld), BF16)
        out.zero_()
        out[:, :d] = _bf(v)
        return 0

    def tok_contrastive_fwd(self, e1, e2, R, n1, n2, d, ld, margin, S, row_loss, loss, st):
        a = _t(e1, (n1, ld), BF16)[:, :d].float()
        b = _t(e2, (n2, ld), BF16)[:, :d].float()
        r = _t(R, (n1, n2), torch.float32)
        ss = (a[:, None, :] - b[None, :, :]).pow(2).sum(-1)
        s = ss.sqrt()
        _t(S, (n1, n2), torch.float32).copy_(s)
        rl = ((1 - r) * (margin - s).clamp_min(0).pow(2) + r * ss).sum(1)
        _t(row_loss, (n1,), torch.float32).copy_(rl)
        _t(loss, (1,), torch.float32).copy_(rl.double().mean().float().reshape(1))
        return 0

    def tok_contrastive_bwd(self, e1, e2, R, S, gscale, n1, n2, d, ld, margin, de1, de2, same, st):
        a = _t(e1, (n1, ld), BF16)[:, :d].float()
        b = _t(e2, (n2, ld), BF16)[:, :d].float()
        r, s = _t(R, (n1, n2), torch.float32), _t(S, (n1, n2), torch.float32)
        g = (_t(gscale, (1,), torch.float32)[0] if gscale else 1.0) / n1
        dls = -2 * (1 - r) * (margin - s).clamp_min(0) + 2 * r * s
        w = torch.where(s > 0, dls / s.clamp_min(1e-30), torch.zeros_like(s))
        diff = a[:, None, :] - b[None, :, :]
        g1 = (w[:, :, None] * diff).sum(1) * g
        g2 = -(w[:, :, None] * diff).sum(0) * g
        o1 = _t(de1, (n1, ld), BF16)
        o1.zero_()
        o1[:, :d] = _bf(g1)
        if same:
            o1[:, :d] = _bf(o1[:, :d].float() + g2)
        else:
            o2 = _t(de2, (n2, ld), BF16)
            o2.zero_()
            o2[:, :d] = _bf(g2)
        return 0

    # ---- multi-resolution glue (HRNet) ----------------------------------------------------------------
    def tok_fuse_sum_relu_fwd(self, t0, s0, t1, s1, t2, s2, t3, s3, n, h, w, c, relu, out, mask, st):
        return self.tok_fuse_sum_affine_relu_fwd(t0, s0, None, None, t1, s1, None, None, t2, s2, None, None, t3, s3, None, None,
                                                 n, h, w, c, relu, out, mask, st)

    def tok_fuse_sum_affine_relu_fwd(self, t0, s0, sc0, sf0, t1, s1, sc1, sf1, t2, s2, sc2, sf2, t3, s3, sc3, sf3, n, h, w, c, relu,
                                     out, mask, st):
        self.calls.append('fuse_sum_relu_fwd')
        acc = torch.zeros(n, h, w, c)
        for tp, sh, sc, sf in ((t0, s0, sc0, sf0), (t1, s1, sc1, sf1), (t2, s2, sc2, sf2), (t3, s3, sc3, sf3)):
            if tp is None:
                continue
            v = _t(tp, (n, h >> sh, w >> sh, c), BF16).float()
            if sc is not None:
                v = v * _t(sc, (c,), torch.float32) + _t(sf, (c,), torch.float32)
            if sh:
                v = v.repeat_interleave(1 << sh, dim=1).repeat_interleave(1 << sh, dim=2)
            acc = acc + v
        if relu:
            acc = acc.clamp_min(0)
        o = acc.to(BF16)
        _t(out, (n, h, w, c), BF16).copy_(o)
        if mask is not None:
            m = n * h * w
            bits = (o.float() > 0).long().reshape(m, c // 8, 8)
            _t(mask, (m, c // 8), torch.uint8).copy_((bits << torch.arange(8)).sum(-1).to(torch.uint8))
        return 0

    def tok_fuse_sum_relu_bwd(self, dout, mask, n, h, w, c, shift, dterm, accumulate, st):
        self.calls.append('fuse_sum_relu_bwd')
        g = _t(dout, (n, h, w, c), BF16).float()
        if mask is not None:
            g = g * self._bits(mask, n * h * w, c).reshape(n, h, w, c)
        f = 1 << shift
        g = g.reshape(n, h // f, f, w // f, f, c).sum((2, 4))
        d = _t(dterm, (n, h // f, w // f, c), BF16)
        d.copy_(_bf(g + d.float()) if accumulate else _bf(g))
        return 0

    def tok_bilinear_fwd(self, src, n, hs, ws, c, ld_src, dst, hd, wd, ld_dst, ch_off, st):
        self.calls.append('bilinear_fwd')
        x = _t(src, (n, hs, ws, ld_src), BF16)[..., :c].float().permute(0, 3, 1, 2)
        y = F.interpolate(x, size=(hd, wd), mode='bilinear', align_corners=False)
        _t(dst, (n, hd, wd, ld_dst), BF16)[..., ch_off:ch_off + c] = _bf(y.permute(0, 2, 3, 1))
        return 0

    def tok_bilinear_bwd(self, ddst, n, hd, wd, ld_dst, ch_off, dsrc, hs, ws, c, ld_src, accumulate, st):
        self.calls.append('bilinear_bwd')
        g = _t(ddst, (n, hd, wd, ld_dst), BF16)[..., ch_off:ch_off + c].float().permute(0, 3, 1, 2)
        with torch.enable_grad():      # called from inside an autograd backward
            x = torch.zeros(n, c, hs, ws, requires_grad=True)
            gx, = torch.autograd.grad(F.interpolate(x, size=(hd, wd), mode='bilinear', align_corners=False), x, g)
        gx = gx.permute(0, 2, 3, 1)
        d = _t(dsrc, (n, hs, ws, ld_src), BF16)
        if accumulate:
            d[..., :c] = _bf(d[..., :c].float() + gx)
        else:
            d[..., :c] = _bf(gx)
        return 0

    def tok_bilinear_bwd_multi(self, ddst, n, hd, wd, c, d1, h1, w1, d2, h2, w2, d3, h3, w3, st):
        for d, hs, ws in ((d1, h1, w1), (d2, h2, w2), (d3, h3, w3)):
            if d is not None:
                self.tok_bilinear_bwd(ddst, n, hd, wd, c, 0, d, hs, ws, c, c, 0, st)
        return 0

    def tok_bilinear_sum_stats_rows(self, n, h, w, c):
        return 1

    def tok_bilinear_sum_stats(self, y0, t1, h1, w1, t2, h2, w2, t3, h3, w3, n, h, w, c, y, stats, st):
        self.calls.append('bilinear_sum_stats')
        acc = _t(y0, (n, h, w, c), BF16).float()
        for t, hs, ws in ((t1, h1, w1), (t2, h2, w2), (t3, h3, w3)):
            if t is None:
                continue
            x = _t(t, (n, hs, ws, c), BF16).float().permute(0, 3, 1, 2)
            acc = acc + F.interpolate(x, size=(h, w), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
        out = _bf(acc)
        _t(y, (n, h, w, c), BF16).copy_(out)
        if stats is not None:
            s_ = _t(stats, (2, 1, c), torch.float32)
            s_.zero_()
            f = out.float().reshape(-1, c)
            s_[0, 0] = f.sum(0)
            s_[1, 0] = (f * f).sum(0)
        return 0

    # ---- token-major transformer units (SwinV2) ----------------------------------------------------------
    def tok_layernorm_fwd(self, x, shortcut, row_scale, rps, gamma, beta, out, mean, rstd, rows, c, ld, eps, st):
        self.calls.append('layernorm_fwd')
        xv = _t(x, (rows, ld), BF16)[:, :c].float()
        mu = xv.mean(1)
        var = ((xv - mu[:, None]) ** 2).mean(1)
        rs = torch.rsqrt(var + eps)
        _t(mean, (rows,), torch.float32).copy_(mu)
        _t(rstd, (rows,), torch.float32).copy_(rs)
        y = (xv - mu[:, None]) * rs[:, None] * _t(gamma, (c,), torch.float32) + _t(beta, (c,), torch.float32)
        if row_scale is not None:
            y = y * _t(row_scale, (rows // rps,), torch.float32).repeat_interleave(rps)[:, None]
        if shortcut is not None:
            y = y + _t(shortcut, (rows, ld), BF16)[:, :c].float()
        o = _t(out, (rows, ld), BF16)
        o.zero_()
        o[:, :c] = _bf(y)
        return 0

    def tok_layernorm_bwd_rows(self, rows, c):
        return 1

    def tok_layernorm_bwd(self, dout, x, mean, rstd, gamma, row_scale, rps, dx, accumulate, partial, rows, c, ld, st):
        self.calls.append('layernorm_bwd')
        xv = _t(x, (rows, ld), BF16)[:, :c].float()
        go = _t(dout, (rows, ld), BF16)[:, :c].float()
        if row_scale is not None:
            go = go * _t(row_scale, (rows // rps,), torch.float32).repeat_interleave(rps)[:, None]
        mu, rs = _t(mean, (rows,), torch.float32), _t(rstd, (rows,), torch.float32)
        xh = (xv - mu[:, None]) * rs[:, None]
        g = go * _t(gamma, (c,), torch.float32)
        d = rs[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
        o = _t(dx, (rows, ld), BF16)
        if accumulate:
            o[:, :c] = _bf(o[:, :c].float() + d)
        else:
            o.zero_()
            o[:, :c] = _bf(d)
        p = _t(partial, (2, 1, c), torch.float32)
        p[0, 0] = (go * xh).sum(0)
        p[1, 0] = go.sum(0)
        return 0

    def tok_colsum_partial_rows(self, m, n_pad):
        return 1

    def tok_colsum_partial(self, dy, m, n_pad, partial, st):
        _t(partial, (1, n_pad), torch.float32).copy_(_t(dy, (m, n_pad), BF16).float().sum(0, keepdim=True))
        return 0

    def tok_colsum_f32(self, src, rows, cols, dst, accumulate, st):
        v = _t(src, (rows, cols), torch.float32).double().sum(0).float()
        d = _t(dst, (cols,), torch.float32)
        d.copy_(d + v if accumulate else v)
        return 0

    def tok_colsum_f32_pair(self, src0, src1, rows, cols, dst0, acc0, dst1, acc1, st):
        self.tok_colsum_f32(src0, rows, cols, dst0, acc0, st)
        return self.tok_colsum_f32(src1, rows, cols, dst1, acc1, st)

    def tok_act_fwd(self, kind, x, out, count, st):
        v = _t(x, (count,), BF16).float()
        _t(out, (count,), BF16).copy_(_bf(v.clamp_min(0) if kind == 0 else F.gelu(v)))
        return 0

    def tok_act_bwd(self, kind, dout, x, dx, accumulate, count, st):
        v = _t(x, (count,), BF16).float()
        g = _t(dout, (count,), BF16).float()
        if kind == 0:
            d = (v > 0).float()
        elif kind == 2:
            d = torch.ones_like(v)
        else:
            d = 0.5 * (1 + torch.erf(v * 0.7071067811865476)) + v * 0.3989422804014327 * torch.exp(-0.5 * v * v)
        o = _t(dx, (count,), BF16)
        o.copy_(_bf(g * d + (o.float() if accumulate else 0)))
        return 0

    @staticmethod
    def _win(t, b, h, w, ws, shift):
        """(B, H, W, X) -> (B*nW, N, X) after roll(-shift) + window_partition."""
        if shift:
            t = torch.roll(t, (-shift, -shift), (1, 2))
        x = t.shape[-1]
        return t.view(b, h // ws, ws, w // ws, ws, x).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, x)

    @staticmethod
    def _unwin(t, b, h, w, ws, shift):
        x = t.shape[-1]
        t = t.view(b, h // ws, w // ws, ws, ws, x).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, x)
        return torch.roll(t, (shift, shift), (1, 2)) if shift else t

    def _attn(self, qkv, b, h, w, c, heads, ws, shift, ld, logit_scale, bias, mask):
        n = ws * ws
        nw = (h // ws) * (w // ws)
        x = _t(qkv, (b, h, w, ld), BF16)[..., :3 * c].float().clone().requires_grad_(True)
        xw = self._win(x, b, h, w, ws, shift).view(-1, n, 3, heads, c // heads).permute(2, 0, 3, 1, 4)
        q, k, v = xw[0], xw[1], xw[2]
        if logit_scale is None:        # plain mode: softmax(q k^T / sqrt(head_dim)) v
            ls = torch.zeros(heads, requires_grad=True)
            bi = torch.zeros(heads, n, n, requires_grad=True)
            attn = (q * (c // heads) ** -0.5) @ k.transpose(-2, -1) + 0 * ls[None, :, None, None] + 0 * bi[None]
        else:
            ls = _t(logit_scale, (heads,), torch.float32).clone().requires_grad_(True)
            bi = _t(bias, (heads, n, n), torch.float32).clone().requires_grad_(True)
            attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
            attn = attn * torch.clamp(ls, max=math.log(100.0)).exp()[None, :, None, None] + bi[None]
        if mask is not None:
            m = _t(mask, (nw, n, n), torch.float32)
            attn = (attn.view(b, nw, heads, n, n) + m[None, :, None]).view(-1, heads, n, n)
        lse = torch.logsumexp(attn, -1)
        o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, n, c)
        return x, ls, bi, self._unwin(o, b, h, w, ws, shift), lse

    def tok_window_attn_fwd(self, qkv, b, h, w, c, heads, ws, shift, ld, logit_scale, bias, mask, out, lse, st):
        self.calls.append('window_attn_fwd')
        with torch.no_grad():
            _, _, _, o, l = self._attn(qkv, b, h, w, c, heads, ws, shift, ld, logit_scale, bias, mask)
        _t(out, (b, h, w, c), BF16).copy_(_bf(o))
        n = ws * ws
        _t(lse, (l.numel(),), torch.float32).copy_(l.reshape(-1))
        return 0

    def tok_window_attn_bwd_rows(self, b, h, w, heads, ws):
        return b * (h // ws) * (w // ws)

    def tok_window_attn_bwd(self, qkv, dout, b, h, w, c, heads, ws, shift, ld, logit_scale, bias, mask, lse, dqkv,
                            ds_scratch, dscale_part, st):
        self.calls.append('window_attn_bwd')
        n = ws * ws
        nw = (h // ws) * (w // ws)
        with torch.enable_grad():
            x, ls, bi, o, _ = self._attn(qkv, b, h, w, c, heads, ws, shift, ld, logit_scale, bias, mask)
            g = _t(dout, (b, h, w, c), BF16).float()
            gx, gls, gbi = torch.autograd.grad(o, (x, ls, bi), g)
        d = _t(dqkv, (b, h, w, ld), BF16)
        d.zero_()
        d[..., :3 * c] = _bf(gx)
        if logit_scale is None:
            return 0
        # the stand-in reports the reduced gradients in row 0 of the scratch buffers (their colsums are what is used)
        sc = _t(ds_scratch, (b * nw, heads, n, n), torch.float32)
        sc.zero_()
        sc[0] = gbi
        dp = _t(dscale_part, (b * nw, heads), torch.float32)
        dp.zero_()
        dp[0] = gls
        return 0

    # ---- DaViT channel attention / pre-norm residual -------------------------------------------------------------------
    @staticmethod
    def _heads(p, ld, rpi, images, heads):
        """bf16 [images*rpi][ld] at p -> float [images, heads, rpi, 32] view of the first heads*32 columns."""
        n = images * rpi
        flat = _t(p, ((n - 1) * ld + heads * 32,), BF16)
        m = torch.as_strided(flat, (n, heads * 32), (ld, 1)).float()
        return m.view(images, rpi, heads, 32).permute(0, 2, 1, 3)

    def tok_chan_gram(self, x, ldx, y, ldy, rpi, images, heads, scale, mode, a_in, out, st):
        xs, ys = self._heads(x, ldx, rpi, images, heads), self._heads(y, ldy, rpi, images, heads)
        g = scale * (xs.transpose(-1, -2) @ ys).reshape(images * heads, 32, 32)
        if mode == 1:
            g = g.softmax(-1)
        elif mode == 2:
            a = _t(a_in, (images * heads, 32, 32), torch.float32)
            g = a * (g - (g * a).sum(-1, keepdim=True))
        _t(out, (images * heads, 32, 32), torch.float32).copy_(g)
        return 0

    def tok_chan_apply(self, x, ldx, m, transposed, scale, rpi, images, heads, out, ldo, st):
        xs = self._heads(x, ldx, rpi, images, heads)                     # [img, h, n, 32]
        mm = _t(m, (images, heads, 32, 32), torch.float32)
        if not transposed:
            mm = mm.transpose(-1, -2)                                    # out = x M^T
        o = scale * (xs @ mm)                                            # [img, h, n, 32]
        n = images * rpi
        flat = _t(out, ((n - 1) * ldo + heads * 32,), BF16)
        dst = torch.as_strided(flat, (n, heads * 32), (ldo, 1))
        dst.copy_(_bf(o.permute(0, 2, 1, 3).reshape(n, heads * 32)))
        return 0

    def tok_scale_rows_add(self, a, b, row_scale, rps, out, accumulate, rows, ld, st):
        bv = _t(b, (rows, ld), BF16).float()
        if row_scale is not None:
            samples = (rows + rps - 1) // rps
            rs = _t(row_scale, (samples,), torch.float32)
            bv = bv * rs.repeat_interleave(rps)[:rows, None]
        if a is not None:
            bv = bv + _t(a, (rows, ld), BF16).float()
        o = _t(out, (rows, ld), BF16)
        if accumulate:
            bv = bv + o.float()
        o.copy_(_bf(bv))
        return 0

    def tok_cpb_bias_fwd(self, table, ld, index, heads, n, bias, st):
        t = _t(table, (int(_t(index, (n * n,), torch.int64).max()) + 1, ld), BF16)
        idx = _t(index, (n * n,), torch.int64)
        b = 16 * torch.sigmoid(t.float()[idx][:, :heads])
        _t(bias, (heads, n, n), torch.float32).copy_(b.t().reshape(heads, n, n))
        return 0

    def tok_cpb_bias_bwd(self, dbias, transposed, table, ld, index, heads, n, rows, dtable, st):
        g = _t(dbias, (heads, n, n), torch.float32)
        if transposed:
            g = g.transpose(1, 2)
        idx = _t(index, (n * n,), torch.int64)
        t = _t(table, (rows, ld), BF16).float()
        acc = torch.zeros(rows, heads).index_add_(0, idx, g.reshape(heads, n * n).t().contiguous())
        s = torch.sigmoid(t[:, :heads])
        d = _t(dtable, (rows, ld), BF16)
        d.zero_()
        d[:, :heads] = _bf(acc * 16 * s * (1 - s))
        return 0

    def tok_patch_merge(self, src, dst, b, h, w, c, inverse, st):
        if not inverse:
            x = _t(src, (b, h, w, c), BF16)
            o = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
            _t(dst, (b, h // 2, w // 2, 4 * c), BF16).copy_(o)
        else:
            x = _t(src, (b, h // 2, w // 2, 4, c), BF16)
            o = _t(dst, (b, h, w, c), BF16)
            o[:, 0::2, 0::2], o[:, 1::2, 0::2], o[:, 0::2, 1::2], o[:, 1::2, 1::2] = x[..., 0, :], x[..., 1, :], \
                x[..., 2, :], x[..., 3, :]
        return 0

    # ---- optimizers ---------------------------------------------------------------------------------
    def tok_sgd_step(self, param, grad, mbuf, shadow, count, lr, momentum, dampening, wd, nesterov, first,
                     maximize, st):
        self.calls.append('sgd_step')
        p, g = _t(param, (count,), torch.float32), _t(grad, (count,), torch.float32)
        d = -g if maximize else g.clone()
        if wd != 0:
            d = d + wd * p
        if momentum != 0:
            b = _t(mbuf, (count,), torch.float32)
            if first:
                b.copy_(d)
            else:
                b.mul_(momentum).add_(d, alpha=1 - dampening)
            d = d + momentum * b if nesterov else b
        p.add_(d, alpha=-lr)
        if shadow is not None:
            _t(shadow, (count,), BF16).copy_(p)
        return 0

    def tok_rmsprop_step(self, param, grad, sq, mbuf, gavg, count, lr, alpha, eps, wd, momentum, centered, maximize, st):
        self.calls.append('rmsprop_step')
        p, g = _t(param, (count,), torch.float32), _t(grad, (count,), torch.float32)
        s_ = _t(sq, (count,), torch.float32)
        d = -g if maximize else g.clone()
        if wd != 0:
            d = d + wd * p
        s_.mul_(alpha).addcmul_(d, d, value=1 - alpha)
        if centered:
            ga = _t(gavg, (count,), torch.float32)
            ga.lerp_(d, 1 - alpha)
            avg = torch.addcmul(s_, ga, ga, value=-1).sqrt_().add_(eps)
        else:
            avg = s_.sqrt().add_(eps)
        if momentum > 0:
            b = _t(mbuf, (count,), torch.float32)
            b.mul_(momentum).addcdiv_(d, avg)
            p.add_(b, alpha=-lr)
        else:
            p.addcdiv_(d, avg, value=-lr)
        return 0

    def tok_adam_step(self, param, grad, m, v, shadow, count, lr, b1, b2, eps, wd, decoupled, step, maximize, st):
        self.calls.append('adam_step')
        p, g = _t(param, (count,), torch.float32), _t(grad, (count,), torch.float32)
        m_, v_ = _t(m, (count,), torch.float32), _t(v, (count,), torch.float32)
        d = -g if maximize else g.clone()
        if wd != 0:
            if decoupled:
                p.mul_(1 - lr * wd)
            else:
                d = d + wd * p
        m_.lerp_(d, 1 - b1)
        v_.mul_(b2).addcmul_(d, d, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = v_.sqrt() / math.sqrt(bc2) + eps
        p.addcdiv_(m_, denom, value=-(lr / bc1))
        if shadow is not None:
            _t(shadow, (count,), BF16).copy_(p)
        return 0

    def tok_adam_step_capturable(self, param, grad, m, v, shadow, count, lr, b1, b2, eps, wd, decoupled, step_dev, maximize, st):
        step = int(_t(step_dev, (1,), torch.int64).item()) + 1
        rc = self.tok_adam_step(param, grad, m, v, shadow, count, lr, b1, b2, eps, wd, decoupled, step, maximize, st)
        self.calls[-1] = 'adam_step_capturable'
        return rc

    def tok_step_advance(self, step_dev, st):
        _t(step_dev, (1,), torch.int64).add_(1)
        return 0

    def tok_fill_f32(self, dst, value, count, st):
        _t(dst, (count,), torch.float32).fill_(value)
        return 0

    def tok_scale_f32(self, dst, f, count, st):
        _t(dst, (count,), torch.float32).mul_(f)
        return 0


# ---- installing the stand-in (tests only; the product has no hook for it) -----------------------------------------------
# torchok_amd refuses host tensors and always asks torch for the current HIP stream.  A host-logic test therefore patches,
# from the OUTSIDE, the three names through which the package reaches the device: `_C._lib` (the loaded library object),
# `stream_ptr` (-> no stream) and `require_device` (-> accepts host tensors) in every module that imported them.
def install(fake=None):
    """Route torchok_amd's native calls to a FakeTok.  Returns an undo token for `uninstall`."""
    import sys

    import torchok_amd  # noqa: F401  (loads the modules that bind stream_ptr / require_device)
    import torchok_amd.dist  # noqa: F401
    import torchok_amd.metrics  # noqa: F401
    import torchok_amd.retrieval  # noqa: F401
    from torchok_amd import _C
    fake = fake if fake is not None else FakeTok()
    undo = [(_C, '_lib', _C._lib)]
    _C._lib = fake

    def no_stream():
        return None

    def any_device(t):
        return None
    for name, mod in list(sys.modules.items()):
        if not name.startswith('torchok_amd') or mod is None:
            continue
        for attr, repl in (('stream_ptr', no_stream), ('require_device', any_device)):
            if attr in getattr(mod, '__dict__', {}):
                undo.append((mod, attr, mod.__dict__[attr]))
                setattr(mod, attr, repl)
    return fake, undo


def uninstall(token):
    _, undo = token
    for mod, attr, old in reversed(undo):
        setattr(mod, attr, old)
