"""CPU restatement of the SwinV2 row, fp32, plain PyTorch.
  * [timm 0.6.13] `timm.models.swin_transformer_v2` pieces the reference imports (`swin.py:18-20`): PatchEmbed,
    PatchMerging, BasicLayer (SwinTransformerBlock, WindowAttention, Mlp, window_partition / window_reverse) and
    `timm.models.layers.{trunc_normal_, to_2tuple, DropPath}` — absent offline, semantics per SURVEY.md App. A.3
  * the wiring of `torchok/models/backbones/swin.py:71-256` is NOT restated here: tests/golden/gen_golden.py and the tests
    run the reference's own SwinTransformerV2 class on top of these pieces; `SwinV2` below is the same wiring for
    boxes without /root/reference (asserted bit-identical to it at fixture generation).
TEST INFRASTRUCTURE ONLY."""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], 'Input image size doesn\'t match model.'
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size[0], window_size[0], W // window_size[1], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size[0], window_size[1], C)


def window_reverse(windows, window_size, img_size):
    H, W = img_size
    B = int(windows.shape[0] / (H * W / window_size[0] / window_size[1]))
    x = windows.view(B, H // window_size[0], W // window_size[1], window_size[0], window_size[1], -1)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0., proj_drop=0.,
                 pretrained_window_size=(0, 0)):
        super().__init__()
        self.dim, self.window_size, self.pretrained_window_size, self.num_heads = dim, window_size, \
            pretrained_window_size, num_heads
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True),
                                     nn.Linear(512, num_heads, bias=False))
        rh = torch.arange(-(window_size[0] - 1), window_size[0], dtype=torch.float32)
        rw = torch.arange(-(window_size[1] - 1), window_size[1], dtype=torch.float32)
        table = torch.stack(torch.meshgrid([rh, rw])).permute(1, 2, 0).contiguous().unsqueeze(0)
        if pretrained_window_size[0] > 0:
            table[:, :, :, 0] /= (pretrained_window_size[0] - 1)
            table[:, :, :, 1] /= (pretrained_window_size[1] - 1)
        else:
            table[:, :, :, 0] /= (window_size[0] - 1)
            table[:, :, :, 1] /= (window_size[1] - 1)
        table *= 8
        table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / math.log2(8)
        self.register_buffer('relative_coords_table', table, persistent=False)
        ch, cw = torch.arange(window_size[0]), torch.arange(window_size[1])
        coords = torch.flatten(torch.stack(torch.meshgrid([ch, cw])), 1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size[0] - 1
        rel[:, :, 1] += window_size[1] - 1
        rel[:, :, 0] *= 2 * window_size[1] - 1
        self.register_buffer('relative_position_index', rel.sum(-1), persistent=False)
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.register_buffer('k_bias', torch.zeros(dim), persistent=False)
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = self.k_bias = self.v_bias = None
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv_bias = None
        if self.q_bias is not None:
            qkv_bias = torch.cat((self.q_bias, self.k_bias, self.v_bias))
        qkv = F.linear(input=x, weight=self.qkv.weight, bias=qkv_bias)
        qkv = qkv.reshape(B_, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
        logit_scale = torch.clamp(self.logit_scale, max=math.log(1. / 0.01)).exp()
        attn = attn * logit_scale
        table = self.cpb_mlp(self.relative_coords_table).view(-1, self.num_heads)
        bias = table[self.relative_position_index.view(-1)].view(
            self.window_size[0] * self.window_size[1], self.window_size[0] * self.window_size[1], -1)
        bias = 16 * torch.sigmoid(bias.permute(2, 0, 1).contiguous())
        attn = attn + bias.unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(B_ // nW, nW, self.num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.num_heads, N, N)
        attn = self.attn_drop(self.softmax(attn))
        x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
        return self.proj_drop(self.proj(x))


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 pretrained_window_size=0):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, to_2tuple(input_resolution), num_heads
        ws, ss = self._calc_window_shift(window_size, shift_size)
        self.window_size, self.shift_size = ws, ss
        self.window_area = ws[0] * ws[1]
        self.mlp_ratio = mlp_ratio
        self.attn = WindowAttention(dim, window_size=to_2tuple(self.window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    attn_drop=attn_drop, proj_drop=drop,
                                    pretrained_window_size=to_2tuple(pretrained_window_size))
        self.norm1 = norm_layer(dim)
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.norm2 = norm_layer(dim)
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        if any(self.shift_size):
            H, W = self.input_resolution
            img_mask = torch.zeros((1, H, W, 1))
            cnt = 0
            for h in (slice(0, -self.window_size[0]), slice(-self.window_size[0], -self.shift_size[0]),
                      slice(-self.shift_size[0], None)):
                for w in (slice(0, -self.window_size[1]), slice(-self.window_size[1], -self.shift_size[1]),
                          slice(-self.shift_size[1], None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mask_windows = window_partition(img_mask, self.window_size).view(-1, self.window_area)
            attn_mask = mask_windows.unsqueeze(1) - mask_windows.unsqueeze(2)
            attn_mask = attn_mask.masked_fill(attn_mask != 0, float(-100.0)).masked_fill(attn_mask == 0, float(0.0))
        else:
            attn_mask = None
        self.register_buffer('attn_mask', attn_mask)

    def _calc_window_shift(self, target_window_size, target_shift_size):
        target_window_size, target_shift_size = to_2tuple(target_window_size), to_2tuple(target_shift_size)
        window_size = [r if r <= w else w for r, w in zip(self.input_resolution, target_window_size)]
        shift_size = [0 if r <= w else s for r, w, s in zip(self.input_resolution, window_size, target_shift_size)]
        return tuple(window_size), tuple(shift_size)

    def _attn(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        has_shift = any(self.shift_size)
        shifted = torch.roll(x, shifts=(-self.shift_size[0], -self.shift_size[1]), dims=(1, 2)) if has_shift else x
        xw = window_partition(shifted, self.window_size).view(-1, self.window_area, C)
        aw = self.attn(xw, mask=self.attn_mask).view(-1, self.window_size[0], self.window_size[1], C)
        shifted = window_reverse(aw, self.window_size, self.input_resolution)
        x = torch.roll(shifted, shifts=self.shift_size, dims=(1, 2)) if has_shift else shifted
        return x.view(B, L, C)

    def forward(self, x):
        x = x + self.drop_path1(self.norm1(self._attn(x)))
        x = x + self.drop_path2(self.norm2(self.mlp(x)))
        return x


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(2 * dim)

    def forward(self, x):
        H, W = self.input_resolution
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        x = torch.cat([x[:, 0::2, 0::2, :], x[:, 1::2, 0::2, :], x[:, 0::2, 1::2, :], x[:, 1::2, 1::2, :]], -1)
        return self.norm(self.reduction(x.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, pretrained_window_size=0):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.grad_checkpointing = False
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer, pretrained_window_size=pretrained_window_size)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None \
            else nn.Identity()

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x)

    def _init_respostnorm(self):
        for blk in self.blocks:
            nn.init.constant_(blk.norm1.bias, 0)
            nn.init.constant_(blk.norm1.weight, 0)
            nn.init.constant_(blk.norm2.bias, 0)
            nn.init.constant_(blk.norm2.weight, 0)


def checkpoint_filter_fn(state_dict, model):
    return state_dict


class _TorchOkBasicLayer(BasicLayer):
    """swin.py:71-81."""

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x), x


class SwinV2(nn.Module):
    """`SwinTransformerV2` of torchok/models/backbones/swin.py:84-256 (same child names), drop rates as arguments."""

    def __init__(self, img_size=256, patch_size=4, in_channels=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4., qkv_bias=True, drop_path_rate=0.1,
                 pretrained_window_sizes=(0, 0, 0, 0)):
        super().__init__()
        self.num_layers = len(depths)
        self.encoder_channels = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        self.out_channels = self.encoder_channels[-1]
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_channels, embed_dim=embed_dim,
                                      norm_layer=nn.LayerNorm)
        g = self.patch_embed.grid_size
        self.input_resolutions = [(g[0] // 2 ** i, g[1] // 2 ** i) for i in range(self.num_layers)]
        self.pos_drop = nn.Dropout(p=0.)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(_TorchOkBasicLayer(
                dim=int(embed_dim * 2 ** i), input_resolution=self.input_resolutions[i], depth=depths[i],
                num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=nn.LayerNorm,
                downsample=PatchMerging if i < self.num_layers - 1 else None,
                pretrained_window_size=pretrained_window_sizes[i]))
        self.feature_norms = nn.ModuleList([nn.LayerNorm(c) for c in self.encoder_channels])
        for m in self.modules():                                   # swin.py:178-189
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        for bly in self.layers:
            bly._init_respostnorm()

    def _to_map(self, x, i):                                       # swin.py:219-238
        x = self.feature_norms[i](x)
        h, w = self.input_resolutions[i]
        return x.view(-1, h, w, self.encoder_channels[i]).permute(0, 3, 1, 2).contiguous()

    def forward_features(self, x):                                 # swin.py:240-249
        feats = [x]
        t = self.pos_drop(self.patch_embed(x))
        for i, layer in enumerate(self.layers):
            t, a = layer(t)
            feats.append(self._to_map(a, i))
        return feats

    def forward(self, x):                                          # swin.py:251-256
        t = self.pos_drop(self.patch_embed(x))
        for layer in self.layers:
            t, _ = layer(t)
        return self._to_map(t, -1)
