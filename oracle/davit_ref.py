"""CPU restatement of the reference's in-tree DaViT: plain PyTorch fp32, each piece citing
torchok/models/backbones/davit.py.  TEST INFRASTRUCTURE ONLY.

Pinned by tests/golden/davit_cls_step.npz: tests/golden/gen_golden.py imports the reference's OWN davit.py (its timm imports —
DropPath, trunc_normal_, to_2tuple, build_model_with_cfg — stubbed by oracle/timm_min.py; everything else in that file
is in-tree code), runs a ClassificationTask training step and asserts this restatement is bit-identical to it
(features, logits, loss, every gradient, the post-step parameters).  Parameter names equal the reference's."""
import itertools

import torch
import torch.nn as nn
import torch.nn.functional as F

from .timm_min import DropPath


class Mlp(nn.Module):
    """modules/bricks/mlp.py:7-41."""

    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class PatchEmbed(nn.Module):
    """:41-86.  Stem (patch 4): 7x7/4 conv then LayerNorm(embed_dim); later stages (patch 2): LayerNorm(in) on the
    tokens, back to a map, 2x2/2 conv (3x3/2 pad 1 when overlapped)."""

    def __init__(self, patch, cin, dim, overlapped=False):
        super().__init__()
        self.patch = patch
        if patch == 4:
            self.proj, self.norm = nn.Conv2d(cin, dim, 7, 4, 3), nn.LayerNorm(dim)
        else:
            k, p = (3, 1) if overlapped else (2, 0)
            self.proj, self.norm = nn.Conv2d(cin, dim, k, 2, p), nn.LayerNorm(cin)

    def forward(self, x, size):
        h, w = size
        tokens = x.dim() == 3
        if tokens:
            b, _, c = x.shape
            x = self.norm(x).reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        _, _, h, w = x.shape
        x = F.pad(x, (0, (-w) % self.patch, 0, (-h) % self.patch))      # :75-78 (no-op when divisible)
        x = self.proj(x)
        size = (x.size(2), x.size(3))
        x = x.flatten(2).transpose(1, 2)
        return (x if tokens else self.norm(x)), size


class ConvPosEnc(nn.Module):
    """:89-128.  With use_act=False the depthwise result is dropped and the input returned (:124-128)."""

    def __init__(self, dim, use_act=False):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)
        self.activation = nn.GELU() if use_act else None

    def forward(self, x, size):
        b, n, c = x.shape
        feat = self.proj(x.transpose(1, 2).view(b, c, *size)).flatten(2).transpose(1, 2)
        return x + self.activation(feat) if self.activation is not None else x


class ChannelAttention(nn.Module):
    """:131-165."""

    def __init__(self, dim, heads, qkv_bias):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.qkv, self.proj = nn.Linear(dim, 3 * dim, bias=qkv_bias), nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        q, k, v = self.qkv(x).reshape(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        attn = ((k * self.scale).transpose(-1, -2) @ v).softmax(dim=-1)
        x = (attn @ q.transpose(-1, -2)).transpose(-1, -2)
        return self.proj(x.transpose(1, 2).reshape(b, n, c))


class WindowAttention(nn.Module):
    """:168-207."""

    def __init__(self, dim, heads, qkv_bias):
        super().__init__()
        self.heads, self.scale = heads, (dim // heads) ** -0.5
        self.qkv, self.proj = nn.Linear(dim, 3 * dim, bias=qkv_bias), nn.Linear(dim, dim)

    def forward(self, x):
        b, n, c = x.shape
        q, k, v = self.qkv(x).reshape(b, n, 3, self.heads, c // self.heads).permute(2, 0, 3, 1, 4)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(b, n, c))


class Block(nn.Module):
    """ChannelBlock :210-271 / SpatialBlock :274-366 (they differ in the attention only)."""

    def __init__(self, kind, dim, heads, window, mlp_ratio, qkv_bias, drop_path, ffn, cpe_act):
        super().__init__()
        self.kind, self.window, self.ffn = kind, window, ffn
        self.cpe = nn.ModuleList([ConvPosEnc(dim, cpe_act), ConvPosEnc(dim, cpe_act)])
        self.norm1 = nn.LayerNorm(dim)
        self.attn = (ChannelAttention if kind == 'channel' else WindowAttention)(dim, heads, qkv_bias)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        if ffn:
            self.norm2 = nn.LayerNorm(dim)
            self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _windows(self, x, size):
        h, w = size
        b, _, c = x.shape
        ws = self.window
        x = F.pad(x.view(b, h, w, c), (0, 0, 0, (ws - w % ws) % ws, 0, (ws - h % ws) % ws))
        hp, wp = x.shape[1:3]
        win = x.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)
        out = self.attn(win).view(b, hp // ws, wp // ws, ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, c)
        return out[:, :h, :w].reshape(b, h * w, c)

    def forward(self, x, size):
        x = self.cpe[0](x, size)
        cur = self.norm1(x)
        cur = self.attn(cur) if self.kind == 'channel' else self._windows(cur, size)
        x = x + self.drop_path(cur)
        x = self.cpe[1](x, size)
        if self.ffn:
            x = x + self.drop_path(self.mlp(self.norm2(x)))
        return x


class DaViT(nn.Module):
    """:402-536."""

    def __init__(self, in_channels=3, patch_size=4, depths=(1, 1, 3, 1), embed_dims=(96, 192, 384, 768),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, drop_path_rate=0.1,
                 overlapped_patch=False, ffn=True, cpe_act=False):
        super().__init__()
        self.architecture = [[i] * d for i, d in enumerate(depths)]
        self.embed_dims, self.num_stages, self.out_channels = embed_dims, len(embed_dims), embed_dims[-1]
        flat = list(itertools.chain(*self.architecture))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 2 * len(flat))]
        self.patch_embeds = nn.ModuleList(
            [PatchEmbed(patch_size if i == 0 else 2, in_channels if i == 0 else embed_dims[i - 1], embed_dims[i],
                        overlapped_patch) for i in range(self.num_stages)])
        blocks, done = [], 0
        for stage in self.architecture:
            blocks.append(nn.ModuleList([nn.Sequential(*[
                Block(kind, embed_dims[item], num_heads[item], window_size, mlp_ratio, qkv_bias,
                      dpr[2 * (lid + done) + aid], ffn, cpe_act) for aid, kind in enumerate(('spatial', 'channel'))])
                for lid, item in enumerate(stage)]))
            done += len(stage)
        self.main_blocks = nn.ModuleList(blocks)
        for i in range(self.num_stages):
            self.add_module(f'norm{i}', nn.LayerNorm(embed_dims[i]))
        for m in self.modules():                                           # init_weights :458-467
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def _stages(self, x):
        x, size = self.patch_embeds[0](x, (x.size(2), x.size(3)))
        feats, sizes = [x], [size]
        for si, stage in enumerate(self.architecture):
            for branch in sorted(set(stage)):
                if branch >= len(feats):
                    x, size = self.patch_embeds[branch](feats[-1], sizes[-1])
                    feats.append(x)
                    sizes.append(size)
            for li, branch in enumerate(stage):
                t = feats[branch]
                for blk in self.main_blocks[si][li]:
                    t = blk(t, sizes[branch])
                feats[branch] = t
        return feats, sizes

    def _map(self, i, t, size):
        return getattr(self, f'norm{i}')(t).view(-1, size[0], size[1], self.embed_dims[i]).permute(0, 3, 1, 2).contiguous()

    def forward_features(self, x):
        feats, sizes = self._stages(x)
        return [x] + [self._map(i, feats[i], sizes[i]) for i in range(self.num_stages)]

    def forward(self, x):
        feats, sizes = self._stages(x)
        return self._map(self.num_stages - 1, feats[-1], sizes[-1])


def davit_t(**kw):
    return DaViT(**dict(dict(embed_dims=(96, 192, 384, 768), depths=(1, 1, 3, 1), num_heads=(3, 6, 12, 24)), **kw))
