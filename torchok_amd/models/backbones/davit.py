"""DaViT backbones (``davit_t`` / ``davit_s`` / ``davit_b``) on the MI355X engine.

Mirrors ``torchok/models/backbones/davit.py`` — the reference's only IN-TREE transformer: ``PatchEmbed`` (:41-86),
``ConvPosEnc`` (:89-128), ``ChannelAttention`` (:131-165), ``WindowAttention`` (:168-207), ``ChannelBlock`` (:210-271),
``SpatialBlock`` (:274-366 with ``_window_partition`` / ``_window_reverse`` :368-399), ``DaViT`` (:402-536) and the
entrypoints (:552-570).  Module / parameter names are the reference's, so its checkpoints load.

Execution: tokens stay one bf16 ``[B*H*W][C]`` matrix in raster order for the whole backbone (a patch embed's
``reshape -> permute -> conv`` is a strided conv on that same NHWC memory); window partition / reverse are index
arithmetic inside the attention kernel; the channel attention streams the token matrix twice
(``engine.transformer.channel_attention``); every Linear runs on the MFMA conv kernels; the backbone is one autograd node.

``ConvPosEnc`` as shipped: with ``use_act=False`` (the default of every entrypoint) its forward computes the depthwise
convolution and then returns the INPUT unchanged (:124-128: the sum is inside ``if self.activation is not None``), so
``cpe.*.proj`` never influences the output and never receives a gradient.  The same holds here — the parameters
exist (checkpoint compatibility), nothing is computed.  With ``cpe_act=True`` the depthwise 3x3 convolution runs
(``engine.transformer.dwconv3x3``) and ``x + GELU(conv(x))`` is what the block sees.
"""
import itertools
import logging
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ... import engine
from ...constructor import BACKBONES
from ...engine import functional as EF
from ...engine import transformer as ET
from ..base import BaseBackbone
from .swin import DropPath, Mlp, _scale_of, to_2tuple, trunc_normal_


class PatchEmbed(nn.Module):
    def __init__(self, patch_size: int = 16, in_channels: int = 3, embed_dim: int = 96, overlapped: bool = False):
        super().__init__()
        patch_size = to_2tuple(patch_size)
        self.patch_size = patch_size
        if patch_size[0] == 4:
            self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=(7, 7), stride=patch_size, padding=(3, 3))
            self.norm = nn.LayerNorm(embed_dim)
        if patch_size[0] == 2:
            kernel = to_2tuple(3 if overlapped else 2)
            pad = to_2tuple(1 if overlapped else 0)
            self.proj = nn.Conv2d(in_channels, embed_dim, kernel_size=kernel, stride=patch_size, padding=pad)
            self.norm = nn.LayerNorm(in_channels)

    def run(self, r, x, batch: int, size: Tuple[int, int]):
        """x: the image (torch NCHW) for the stem, token rows (B*H*W, C) afterwards.  -> (token rows, (H', W'))."""
        h, w = size
        if h % self.patch_size[0] or w % self.patch_size[1]:
            raise NotImplementedError('torchok_amd DaViT PatchEmbed: feature maps divisible by the patch size (no padding)')
        if isinstance(x, torch.Tensor):                       # stem: conv -> flatten -> norm (:82-85)
            t = r.input(x, c_pad_to=4 if x.shape[1] <= 4 else 8)
            t = EF.conv_bn_act(r, t, self.proj, None, False, None)
            n, gh, gw, cp = t.shape
            return ET.layer_norm(r, ET.reshape(r, t, (n * gh * gw, cp)), self.norm), (gh, gw)
        t = ET.layer_norm(r, x, self.norm)                    # :68-70: norm -> (B, H, W, C) -> conv
        t = ET.reshape(r, t, (batch, h, w, t.cp))
        t = EF.conv_bn_act(r, t, self.proj, None, False, None)
        n, gh, gw, cp = t.shape
        return ET.reshape(r, t, (n * gh * gw, cp)), (gh, gw)


class ConvPosEnc(nn.Module):
    """davit.py:89-128.  Without activation: parameters only (see the module docstring).  With it:
    x + GELU(depthwise3x3(x)) on the (B, H, W, C) view of the tokens."""

    def __init__(self, dim: int, kernel_size: int = 3, use_act: bool = False, norm_layer: Optional[str] = None):
        super().__init__()
        if kernel_size != 3 or norm_layer is not None:
            raise NotImplementedError('torchok_amd DaViT ConvPosEnc: kernel 3, no normalisation (what the blocks use)')
        self.proj = nn.Conv2d(dim, dim, kernel_size, 1, kernel_size // 2, groups=dim)
        self.norm_layer = norm_layer
        self.activation = nn.GELU() if use_act else None

    def run(self, r, x, batch: int, size: Tuple[int, int]):
        if self.activation is None:
            return x                                            # :124-128: the sum sits inside `if self.activation`
        h, w = size
        feat = ET.dwconv3x3(r, ET.reshape(r, x, (batch, h, w, x.cp)), self.proj)
        feat = ET.activation(r, ET.reshape(r, feat, (batch * h * w, x.cp)), ET.GELU)
        return ET.residual_add(r, x, feat)


class ChannelAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def run(self, r, x, batch: int, tokens: int):
        qkv = ET.linear_module(r, x, self.qkv)
        a = ET.channel_attention(r, qkv, batch, tokens, self.num_heads)
        return ET.linear_module(r, a, self.proj)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.softmax = nn.Softmax(dim=-1)

    def run(self, r, x, batch: int, size: Tuple[int, int], window: int):
        h, w = size
        qkv = ET.linear_module(r, x, self.qkv)
        a = ET.window_attention(r, qkv, (batch, h, w, self.dim, self.num_heads, window, 0))
        return ET.linear_module(r, a, self.proj)


class _Block(nn.Module):
    """x = x + drop_path(attn(norm1(x)));  x = x + drop_path(mlp(norm2(x)))  with the two (inert) ConvPosEnc in between."""

    def _tail(self, r, x, cur, batch: int, size: Tuple[int, int]):
        dev = x.data.device
        tokens = size[0] * size[1]
        x = ET.residual_add(r, x, cur, _scale_of(self.drop_path, batch, dev), tokens)
        x = self.cpe[1].run(r, x, batch, size)
        if self.ffn:
            m = self.mlp.run(r, ET.layer_norm(r, x, self.norm2))
            x = ET.residual_add(r, x, m, _scale_of(self.drop_path, batch, dev), tokens)
        return x


class ChannelBlock(_Block):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4., qkv_bias: bool = False, drop_path: float = 0.,
                 act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = nn.LayerNorm, ffn: bool = True,
                 cpe_act: bool = False):
        super().__init__()
        self.cpe = nn.ModuleList([ConvPosEnc(dim=dim, kernel_size=3, use_act=cpe_act),
                                  ConvPosEnc(dim=dim, kernel_size=3, use_act=cpe_act)])
        self.ffn = ffn
        self.norm1 = norm_layer(dim)
        self.attn = ChannelAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        if self.ffn:
            self.norm2 = norm_layer(dim)
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def run(self, r, x, batch: int, size: Tuple[int, int]):
        tokens = size[0] * size[1]
        x = self.cpe[0].run(r, x, batch, size)
        cur = self.attn.run(r, ET.layer_norm(r, x, self.norm1), batch, tokens)
        return self._tail(r, x, cur, batch, size)


class SpatialBlock(_Block):
    def __init__(self, dim: int, num_heads: int, window_size: int = 7, mlp_ratio: float = 4., qkv_bias: bool = True,
                 drop_path: float = 0., act_layer: nn.Module = nn.GELU, norm_layer: nn.Module = nn.LayerNorm,
                 ffn: bool = True, cpe_act: bool = False):
        super().__init__()
        self.dim, self.ffn, self.num_heads, self.window_size, self.mlp_ratio = dim, ffn, num_heads, window_size, mlp_ratio
        self.cpe = nn.ModuleList([ConvPosEnc(dim=dim, kernel_size=3, use_act=cpe_act),
                                  ConvPosEnc(dim=dim, kernel_size=3, use_act=cpe_act)])
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        if self.ffn:
            self.norm2 = norm_layer(dim)
            self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def run(self, r, x, batch: int, size: Tuple[int, int]):
        h, w = size
        if h % self.window_size or w % self.window_size:
            raise NotImplementedError(f'torchok_amd DaViT SpatialBlock: {h}x{w} tokens are not a multiple of the window '
                                      f'{self.window_size} (the zero-padded windows of davit.py:336-340 are not provided)')
        x = self.cpe[0].run(r, x, batch, size)
        cur = self.attn.run(r, ET.layer_norm(r, x, self.norm1), batch, size, self.window_size)
        return self._tail(r, x, cur, batch, size)


class DaViT(BaseBackbone):
    """Dual Attention Transformer (davit.py:402-536)."""

    def __init__(self, img_size: int = 224, in_channels: int = 3, patch_size: int = 4, depths=(1, 1, 3, 1),
                 embed_dims: Tuple[int] = (64, 128, 192, 256), num_heads: Tuple[int] = (3, 6, 12, 24),
                 window_size: int = 7, mlp_ratio: float = 4., qkv_bias: bool = True, drop_path_rate: float = 0.1,
                 norm_layer: nn.Module = nn.LayerNorm, overlapped_patch: bool = False, ffn: bool = True,
                 cpe_act: bool = False):
        super().__init__(in_channels, embed_dims[-1])
        if norm_layer is not nn.LayerNorm:
            raise NotImplementedError('torchok_amd DaViT: LayerNorm')
        if any(d != 32 * h for d, h in zip(embed_dims, num_heads)):
            raise NotImplementedError('torchok_amd DaViT: head_dim 32 (davit_t / davit_s / davit_b)')
        self.img_size = img_size
        architecture = [[index] * item for index, item in enumerate(depths)]
        self.attention_types = ('spatial', 'channel')
        self.architecture = architecture
        self.embed_dims = embed_dims
        self.num_heads = num_heads
        self.num_stages = len(self.embed_dims)
        self._out_encoder_channels = embed_dims
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, 2 * len(list(itertools.chain(*self.architecture))))]
        self.patch_embeds = nn.ModuleList([
            PatchEmbed(patch_size=patch_size if i == 0 else 2, in_channels=in_channels if i == 0 else self.embed_dims[i - 1],
                       embed_dim=self.embed_dims[i], overlapped=overlapped_patch)
            for i in range(self.num_stages)])
        main_blocks = []
        for block_id, block_param in enumerate(self.architecture):
            layer_offset_id = len(list(itertools.chain(*self.architecture[:block_id])))
            block = nn.ModuleList([
                nn.Sequential(*[
                    ChannelBlock(dim=self.embed_dims[item], num_heads=self.num_heads[item], mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, drop_path=dpr[2 * (layer_id + layer_offset_id) + attention_id],
                                 norm_layer=norm_layer, ffn=ffn, cpe_act=cpe_act)
                    if attention_type == 'channel' else
                    SpatialBlock(dim=self.embed_dims[item], num_heads=self.num_heads[item], mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, drop_path=dpr[2 * (layer_id + layer_offset_id) + attention_id],
                                 norm_layer=norm_layer, ffn=ffn, cpe_act=cpe_act, window_size=window_size)
                    for attention_id, attention_type in enumerate(self.attention_types)])
                for layer_id, item in enumerate(block_param)])
            main_blocks.append(block)
        self.main_blocks = nn.ModuleList(main_blocks)
        for i_layer in range(self.num_stages):
            self.add_module(f'norm{i_layer}', norm_layer(self.embed_dims[i_layer]))
        self.init_weights()

    @torch.jit.ignore
    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def _forward_stages(self, r, image: torch.Tensor):
        """davit.py:465-482."""
        batch = image.shape[0]
        x, size = self.patch_embeds[0].run(r, image, batch, (image.size(2), image.size(3)))
        features, sizes, branches = [x], [size], [0]
        for block_index, block_param in enumerate(self.architecture):
            for branch_id in sorted(set(block_param)):
                if branch_id not in branches:
                    x, size = self.patch_embeds[branch_id].run(r, features[-1], batch, sizes[-1])
                    features.append(x)
                    sizes.append(size)
                    branches.append(branch_id)
            for layer_index, branch_id in enumerate(block_param):
                t = features[branch_id]
                for blk in self.main_blocks[block_index][layer_index]:
                    t = blk.run(r, t, batch, sizes[branch_id])
                features[branch_id] = t
        return features, sizes

    def _to_map(self, r, t, i: int, size):
        t = ET.layer_norm(r, t, getattr(self, f'norm{i}'))
        return ET.reshape(r, t, (-1, size[0], size[1], t.cp))

    def forward_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        with engine.region() as r:
            features, sizes = self._forward_stages(r, x)
            outs = r.output(*[self._to_map(r, features[i], i, sizes[i]) for i in range(self.num_stages)])
        return [x] + list(outs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with engine.region() as r:
            features, sizes = self._forward_stages(r, x)
            last = self.num_stages - 1
            return r.output(self._to_map(r, features[last], last, sizes[last]))

    def get_stages(self, stage: int) -> nn.Module:
        logging.warning('DaViT does not support `get_stages`. Return the whole model')
        return self


def _create_davit(variant: str, pretrained: bool = False, **kwargs):
    for k in ('num_classes', 'global_pool', 'in_chans'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a download (no network here); pass '
                           f'pretrained=false and use task.load_checkpoint for local checkpoints')
    return DaViT(**kwargs)


@BACKBONES.register_class
def davit_t(pretrained: bool = False, **kwargs):
    return _create_davit('davit_t', pretrained, **dict(embed_dims=(96, 192, 384, 768), depths=(1, 1, 3, 1),
                                                       num_heads=(3, 6, 12, 24), **kwargs))


@BACKBONES.register_class
def davit_s(pretrained: bool = False, **kwargs):
    return _create_davit('davit_s', pretrained, **dict(embed_dims=(96, 192, 384, 768), depths=(1, 1, 9, 1),
                                                       num_heads=(3, 6, 12, 24), **kwargs))


@BACKBONES.register_class
def davit_b(pretrained: bool = False, **kwargs):
    return _create_davit('davit_b', pretrained, **dict(embed_dims=(128, 256, 512, 1024), depths=(1, 1, 9, 1),
                                                       num_heads=(4, 8, 16, 32), **kwargs))
