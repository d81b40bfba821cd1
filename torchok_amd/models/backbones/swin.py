"""SwinV2 backbones on the MI355X engine.

Mirrors the reference wiring ``torchok/models/backbones/swin.py``: ``BasicLayer.forward`` (:71-81, returns
``(downsample(x), x)``), ``SwinTransformerV2.__init__`` (:108-176), ``init_weights`` (:178-189),
``no_weight_decay`` (:191-202), ``_normalize_with_bhwc_reshape`` (:219-238), ``forward_features`` (:240-249),
``forward`` (:251-256), ``load_state_dict`` (:258-264), ``get_stages`` (:266-276) and the entrypoints (:286-403),
with the [timm 0.6.13] ``swin_transformer_v2`` pieces (PatchEmbed, WindowAttention, SwinTransformerBlock,
PatchMerging, BasicLayer, Mlp, DropPath) restated here per SURVEY.md App. A.3.  Module / parameter / buffer
names are timm's, so reference checkpoints load.

Execution: tokens stay one bf16 ``[B*H*W][C]`` matrix for the whole backbone (the reference's
``view / roll / window_partition / permute / contiguous`` round trips are index arithmetic inside the attention
kernel); every Linear runs on the MFMA conv kernels; ``x + drop_path(norm(f(x)))`` is one fused
LayerNorm-residual kernel; the whole backbone is a single autograd node.
"""
import math
from typing import Any, List, Mapping

import torch
import torch.nn as nn

from ... import engine
from ...constructor import BACKBONES
from ...engine import functional as EF
from ...engine import transformer as ET
from ..base import BaseBackbone


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Stochastic depth: the per-sample keep/scale vector is drawn here and applied inside the fused
    LayerNorm-residual kernel (``row_scale``)."""

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def sample_scale(self, batch: int, device):
        if self.drop_prob == 0. or not self.training:
            return None
        pre, self._drawn = getattr(self, '_drawn', None), None
        if pre is not None and pre.shape[0] == batch and pre.device == device:
            return pre                                   # drawn with the whole step's vectors (draw_drop_scales)
        keep = 1 - self.drop_prob
        mask = torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask


_drop_consts = {}     # (device, drop probabilities ...) -> [2][n][1] fp32: keep probability, scale of a kept sample


def draw_drop_scales(paths, batch: int, device):
    """The keep/scale vectors of every active DropPath of a model in three launches instead of two per module (44 small
    launches on the dependent chain of a SwinV2-T step): one uniform draw for all of them, row i kept where u < keep_i and
    scaled by 1 / keep_i.  Same distribution as the per-module bernoulli_; each module's next `sample_scale` returns its
    row."""
    live = [p for p in paths if isinstance(p, DropPath) and p.training and p.drop_prob > 0.]
    if not live:
        return
    key = (device,) + tuple((p.drop_prob, p.scale_by_keep) for p in live)
    kv = _drop_consts.get(key)
    if kv is None:
        keep = [[1. - p.drop_prob] for p in live]
        val = [[1. / (1. - p.drop_prob) if (p.scale_by_keep and p.drop_prob < 1.) else 1.] for p in live]
        kv = _drop_consts[key] = torch.tensor([keep, val], dtype=torch.float32).to(device)
    u = torch.rand((len(live), batch), dtype=torch.float32, device=device)
    scales = torch.where(u < kv[0], kv[1], 0.)
    for i, p in enumerate(live):
        p._drawn = scales[i]


def _scale_of(dp, batch, device):
    return dp.sample_scale(batch, device) if isinstance(dp, DropPath) else None


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if act_layer is not nn.GELU or drop != 0.:
            raise NotImplementedError('torchok_amd Mlp: GELU, no dropout')
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop2 = nn.Dropout(drop)

    def run(self, r, x):
        return ET.mlp_module(r, x, self.fc1, self.fc2)         # one launch where the geometry is served (csrc/mlp_fused.hip)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size, self.patch_size = to_2tuple(img_size), to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def run(self, r, image: torch.Tensor):
        _, _, H, W = image.shape
        if H != self.img_size[0] or W != self.img_size[1]:
            raise AssertionError("Input image size doesn't match model.")
        t = r.input(image, c_pad_to=4 if image.shape[1] <= 4 else 8)
        t = EF.conv_bn_act(r, t, self.proj, None, False, None)          # (B, H/4, W/4, D), bias in the epilogue
        n, gh, gw, cp = t.shape
        t = ET.reshape(r, t, (n * gh * gw, cp))
        if isinstance(self.norm, nn.LayerNorm):
            t = ET.layer_norm(r, t, self.norm)
        return t


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, attn_drop=0., proj_drop=0.,
                 pretrained_window_size=(0, 0)):
        super().__init__()
        if attn_drop != 0. or proj_drop != 0.:
            raise NotImplementedError('torchok_amd WindowAttention: no attention / projection dropout')
        self.dim, self.window_size, self.pretrained_window_size, self.num_heads = dim, tuple(window_size), \
            tuple(pretrained_window_size), num_heads
        if self.window_size[0] != self.window_size[1]:
            raise NotImplementedError('torchok_amd WindowAttention: square windows')
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(2, 512, bias=True), nn.ReLU(inplace=True),
                                     nn.Linear(512, num_heads, bias=False))
        ws = self.window_size
        rh = torch.arange(-(ws[0] - 1), ws[0], dtype=torch.float32)
        rw = torch.arange(-(ws[1] - 1), ws[1], dtype=torch.float32)
        table = torch.stack(torch.meshgrid([rh, rw], indexing='ij')).permute(1, 2, 0).contiguous().unsqueeze(0)
        den = self.pretrained_window_size if self.pretrained_window_size[0] > 0 else ws
        table[:, :, :, 0] /= (den[0] - 1)
        table[:, :, :, 1] /= (den[1] - 1)
        table *= 8
        table = torch.sign(table) * torch.log2(torch.abs(table) + 1.0) / math.log2(8)
        self.register_buffer('relative_coords_table', table, persistent=False)
        coords = torch.flatten(torch.stack(torch.meshgrid([torch.arange(ws[0]), torch.arange(ws[1])], indexing='ij')), 1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws[0] - 1
        rel[:, :, 1] += ws[1] - 1
        rel[:, :, 0] *= 2 * ws[1] - 1
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

    def prepare(self, r, stream: int = 1):
        """The part of the forward that reads parameters only — continuous position bias (cpb_mlp over the (2w-1)^2
        coordinate table, 16 * sigmoid, index gather) and the concatenated qkv bias: seven small launches, recorded on
        branch stream `stream` so that they (and their backward) stay off the dependent chain of the main stream.  `run`
        picks the result up; a model calls this one block AHEAD of `run` (SwinTransformerV2._run)."""
        with r.branch(stream, fork=False) as br:
            tab = self.relative_coords_table
            pad = getattr(self, '_table_pad', None)
            if pad is None or pad.device != tab.device:
                # bf16 rows padded to 8 channels once (the table is a constant buffer): Region.input takes the view as is
                pad = torch.zeros((tab.numel() // 2, 8), dtype=torch.bfloat16, device=tab.device)
                pad[:, :2] = tab.view(-1, 2)
                engine.mark_padded(pad)
                self._table_pad = pad
            table_in = r.input(pad[:, :2])
            t = ET.linear_module(r, table_in, self.cpb_mlp[0])
            t = ET.activation(r, t, ET.RELU)
            t = ET.linear_module(r, t, self.cpb_mlp[2])
            n = self.window_size[0] * self.window_size[0]
            bias, bias_node = ET.cpb_bias(r, t, self.relative_position_index, self.num_heads, n)
            qkv_bias = None
            if self.q_bias is not None:
                qkv_bias = torch.cat((self.q_bias.detach(), self.k_bias, self.v_bias.detach()))
            self._prepared = br.publish(_Prepared(bias, bias_node, qkv_bias))

    def run(self, r, x, batch: int, res, shift: int, mask):
        """x: token rows (B*H*W, C) in raster order -> attention + proj, same order."""
        H, W = res
        ws = self.window_size[0]
        if getattr(self, '_prepared', None) is None:
            self.prepare(r)
        pre, self._prepared = self._prepared, None
        engine.await_ready(pre)            # (+ record_stream: bias / qkv_bias live in the branch stream's pool)
        if pre.qkv_bias is not None:
            qkv = ET.linear_op(r, x, self.qkv.weight, pre.qkv_bias, [(self.q_bias, 0), (self.v_bias, 2 * self.dim)])
        else:
            qkv = ET.linear_op(r, x, self.qkv.weight)
        a = ET.window_attention(r, qkv, (batch, H, W, self.dim, self.num_heads, ws, shift), self.logit_scale, pre.bias,
                                pre.bias_node, mask)
        return ET.linear_module(r, a, self.proj)


class _Prepared:
    __slots__ = ('bias', 'bias_node', 'qkv_bias', 'ready')

    def __init__(self, bias, bias_node, qkv_bias):
        self.bias, self.bias_node, self.qkv_bias, self.ready = bias, bias_node, qkv_bias, None

    @property
    def tensors(self):
        return (self.bias, self.qkv_bias)


def window_partition(x, window_size):
    B, H, W, C = x.shape
    x = x.view(B, H // window_size[0], window_size[0], W // window_size[1], window_size[1], C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, window_size[0], window_size[1], C)


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
            for h in (slice(0, -ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
                for w in (slice(0, -ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
                    img_mask[:, h, w, :] = cnt
                    cnt += 1
            mask_windows = window_partition(img_mask, ws).view(-1, self.window_area)
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

    def run(self, r, x, batch: int):
        H, W = self.input_resolution
        if self.shift_size[0] != self.shift_size[1] or self.window_size[0] != self.window_size[1]:
            raise NotImplementedError('torchok_amd SwinTransformerBlock: square feature maps / windows')
        L = H * W
        dev = x.data.device
        a = self.attn.run(r, x, batch, (H, W), self.shift_size[0], self.attn_mask)
        x = ET.layer_norm(r, a, self.norm1, shortcut=x, row_scale=_scale_of(self.drop_path1, batch, dev),
                          rows_per_sample=L)                       # x + drop_path1(norm1(attn(x)))
        m = self.mlp.run(r, x)
        return ET.layer_norm(r, m, self.norm2, shortcut=x, row_scale=_scale_of(self.drop_path2, batch, dev),
                             rows_per_sample=L)                    # x + drop_path2(norm2(mlp(x)))


class PatchMerging(nn.Module):
    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(2 * dim)

    def run(self, r, x, batch: int):
        H, W = self.input_resolution
        if H % 2 or W % 2:
            raise AssertionError(f'x size ({H}*{W}) are not even.')
        t = ET.patch_merge(r, x, batch, H, W)
        t = ET.linear_module(r, t, self.reduction)
        return ET.layer_norm(r, t, self.norm)


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

    def run(self, r, x, batch: int, ahead=None):
        """swin.py:71-81: (downsample(x), x).  `ahead`: iterator over the model's blocks whose attention units have not
        been prepared yet (WindowAttention.prepare runs two blocks ahead of its consumer)."""
        for blk in self.blocks:
            nxt = next(ahead, None) if ahead is not None else None
            if nxt is not None:
                nxt.attn.prepare(r)
            x = blk.run(r, x, batch)
        down = self.downsample.run(r, x, batch) if isinstance(self.downsample, PatchMerging) else x
        return down, x

    def _init_respostnorm(self):
        for blk in self.blocks:
            nn.init.constant_(blk.norm1.bias, 0)
            nn.init.constant_(blk.norm1.weight, 0)
            nn.init.constant_(blk.norm2.bias, 0)
            nn.init.constant_(blk.norm2.weight, 0)


class SwinTransformerV2(BaseBackbone):
    def __init__(self, img_size: int = 256, patch_size: int = 4, in_channels: int = 3, embed_dim: int = 96,
                 depths: List[int] = (2, 2, 6, 2), num_heads: List[int] = (3, 6, 12, 24), window_size: int = 7,
                 mlp_ratio: float = 4., qkv_bias: bool = True, drop_rate: float = 0., attn_drop_rate: float = 0.,
                 drop_path_rate: float = 0.1, norm_layer: nn.Module = nn.LayerNorm, ape: bool = False,
                 patch_norm: bool = True, pretrained_window_sizes: List[int] = (0, 0, 0, 0),
                 load_attn_mask: bool = True):
        super().__init__(in_channels=in_channels)
        if norm_layer is not nn.LayerNorm or ape or drop_rate != 0.:
            raise NotImplementedError('torchok_amd SwinTransformerV2: LayerNorm, no absolute position embedding, '
                                      'drop_rate = 0')
        self.img_size = to_2tuple(img_size)
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.ape = ape
        self.patch_norm = patch_norm
        self.encoder_channels = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        self._out_channels = self.encoder_channels[-1]
        self._out_encoder_channels = self.encoder_channels
        self.load_attn_mask = load_attn_mask
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_channels, embed_dim=embed_dim,
                                      norm_layer=norm_layer if self.patch_norm else None)
        g = self.patch_embed.grid_size
        self.input_resolutions = [(g[0] // (2 ** i), g[1] // (2 ** i)) for i in range(self.num_layers)]
        self.patches_resolution = g
        self.absolute_pos_embed = None
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), input_resolution=self.input_resolutions[i], depth=depths[i],
                num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                norm_layer=norm_layer, downsample=PatchMerging if (i < self.num_layers - 1) else None,
                pretrained_window_size=pretrained_window_sizes[i]))
        self.feature_norms = nn.ModuleList([norm_layer(chs) for chs in self.encoder_channels])
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        for bly in self.layers:
            bly._init_respostnorm()

    def no_weight_decay(self) -> List[str]:
        nod = ['absolute_pos_embed']
        for n, m in self.named_modules():
            if any([kw in n for kw in ('cpb_mlp', 'logit_scale', 'relative_position_bias_table')]):
                nod.append(n)
        return nod

    def _to_map(self, r, x, layer_number: int):
        """_normalize_with_bhwc_reshape: feature_norms[i], then (B*L, C) -> (B, H, W, C) (returned as an NCHW view)."""
        x = ET.layer_norm(r, x, self.feature_norms[layer_number])
        h, w = self.input_resolutions[layer_number]
        return ET.reshape(r, x, (-1, h, w, x.cp))

    def _run(self, r, image: torch.Tensor, all_features: bool):
        batch = image.shape[0]
        blocks = [blk for layer in self.layers for blk in layer.blocks]
        if self.training:
            draw_drop_scales([p for blk in blocks for p in (blk.drop_path1, blk.drop_path2)], batch, image.device)
        for blk in blocks:
            blk.attn._prepared = None
        # the parameter-only part of every attention unit runs one block ahead of its consumer, on a branch stream
        r.device = image.device
        ahead = iter(blocks)
        for blk in blocks[:2]:
            next(ahead).attn.prepare(r)
        t = self.patch_embed.run(r, image)
        feats = []
        for i, layer in enumerate(self.layers):
            t, a = layer.run(r, t, batch, ahead)
            if all_features or i == self.num_layers - 1:
                feats.append(self._to_map(r, a, i))
        return feats

    def forward_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        with engine.region() as r:
            outs = r.output(*self._run(r, x, True))
        return [x] + list(outs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with engine.region() as r:
            return r.output(self._run(r, x, False)[-1])

    def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True):
        if not self.load_attn_mask:
            state_dict = dict(state_dict)
            for k in list(state_dict.keys()):
                if 'attn_mask' in k:
                    state_dict.pop(k)
        return super().load_state_dict(state_dict, strict)

    def get_stages(self, stage: int) -> nn.Module:
        output = [self.patch_embed, self.pos_drop]
        return nn.ModuleList(output + list(self.layers[:stage]))


def _create_swin_transformer_v2(variant, pretrained=False, **kwargs):
    for k in ('num_classes', 'global_pool', 'in_chans'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a download (no network here); pass '
                           f'pretrained=false and use task.load_checkpoint for local checkpoints')
    return SwinTransformerV2(**kwargs)


_VARIANTS = {
    'swinv2_custom': dict(),
    'swinv2_tiny_window16_256': dict(window_size=16, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
    'swinv2_tiny_window8_256': dict(window_size=8, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24)),
    'swinv2_small_window16_256': dict(window_size=16, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    'swinv2_small_window8_256': dict(window_size=8, embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24)),
    'swinv2_base_window16_256': dict(window_size=16, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    'swinv2_base_window8_256': dict(window_size=8, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    'swinv2_base_window12_192_22k': dict(window_size=12, embed_dim=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32)),
    'swinv2_base_window12to16_192to256_22kft1k': dict(window_size=16, embed_dim=128, depths=(2, 2, 18, 2),
                                                      num_heads=(4, 8, 16, 32), pretrained_window_sizes=(12, 12, 12, 6)),
    'swinv2_base_window12to24_192to384_22kft1k': dict(window_size=24, embed_dim=128, depths=(2, 2, 18, 2),
                                                      num_heads=(4, 8, 16, 32), pretrained_window_sizes=(12, 12, 12, 6)),
    'swinv2_large_window12_192_22k': dict(window_size=12, embed_dim=192, depths=(2, 2, 18, 2), num_heads=(6, 12, 24, 48)),
    'swinv2_large_window12to16_192to256_22kft1k': dict(window_size=16, embed_dim=192, depths=(2, 2, 18, 2),
                                                       num_heads=(6, 12, 24, 48), pretrained_window_sizes=(12, 12, 12, 6)),
    'swinv2_large_window12to24_192to384_22kft1k': dict(window_size=24, embed_dim=192, depths=(2, 2, 18, 2),
                                                       num_heads=(6, 12, 24, 48), pretrained_window_sizes=(12, 12, 12, 6)),
}


def _register(variant, defaults):
    def entry(pretrained=False, **kwargs):
        return _create_swin_transformer_v2(variant, pretrained=pretrained, **dict(defaults, **kwargs))
    entry.__name__ = variant
    return BACKBONES.register_class(entry)


for _name, _kw in _VARIANTS.items():
    globals()[_name] = _register(_name, _kw)
