"""ResNet backbones on the MI355X engine.

Mirrors the reference wiring ``torchok/models/backbones/resnet.py``: ``make_blocks`` (:363-405),
``ResNet.__init__`` (:462-526), ``init_weights`` (:529-539), ``forward`` (:541-551), ``get_stages``
(:553-563) and the ``resnet18/34/50/101/152`` entrypoints (:589-594, :648-653 ...), with the
[timm 0.6.13] ``BasicBlock`` / ``Bottleneck`` / ``downsample_conv`` semantics restated here
(SURVEY.md App. A.1).  Module / parameter names are those of timm, so reference checkpoints load.

Execution differs completely: each ``conv -> bn -> (+shortcut) -> relu`` group is ONE engine unit
(implicit-GEMM MFMA conv with BatchNorm statistics in its epilogue, one fused normalise/add/ReLU
pass), activations stay NHWC bf16 in HBM between units, and the whole backbone is a single
autograd node.  The nn.Conv2d / nn.BatchNorm2d children are parameter containers only.

Only the plain (v1.5) family is built: no SE / anti-aliasing / drop-block / avg-down / deep stem
(plus the other plain-architecture entrypoints at the end of the file; the d/s/t-stem, grouped, SE/ECA, blur-pool and
RS variants of the reference are outside the hot-path scope).
"""
import math
from typing import List

import torch
import torch.nn as nn

from ... import engine
from ...constructor import BACKBONES
from ...engine import functional as EF
from ..base import BaseBackbone


def get_padding(kernel_size: int, stride: int, dilation: int = 1) -> int:
    return ((stride - 1) + dilation * (kernel_size - 1)) // 2


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64,
                 reduce_first=1, dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d,
                 attn_layer=None, aa_layer=None, drop_block=None, drop_path=None):
        super().__init__()
        assert cardinality == 1 and base_width == 64, 'BasicBlock only supports cardinality=1, base_width=64'
        _unsupported(dilation=dilation != 1, attn_layer=attn_layer, aa_layer=aa_layer, drop_block=drop_block,
                     drop_path=drop_path)
        first_planes = planes // reduce_first
        outplanes = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = act_layer(inplace=True)
        self.conv2 = nn.Conv2d(first_planes, outplanes, kernel_size=3, padding=1, bias=False)
        self.bn2 = norm_layer(outplanes)
        self.act2 = act_layer(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def zero_init_last(self):
        nn.init.zeros_(self.bn2.weight)

    def forward(self, x):
        r = engine.current_region()
        shortcut = self._shortcut(r, x)
        y = EF.conv_bn_act(r, x, self.conv1, self.bn1, relu=True)
        return EF.conv_bn_act(r, y, self.conv2, self.bn2, relu=True, shortcut=shortcut)

    def _shortcut(self, r, x):
        return _shortcut_branch(r, x, self.downsample)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64,
                 reduce_first=1, dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d,
                 attn_layer=None, aa_layer=None, drop_block=None, drop_path=None):
        super().__init__()
        _unsupported(cardinality=cardinality != 1, dilation=dilation != 1, attn_layer=attn_layer,
                     aa_layer=aa_layer, drop_block=drop_block, drop_path=drop_path)
        width = int(math.floor(planes * (base_width / 64)) * cardinality)
        first_planes = width // reduce_first
        outplanes = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = act_layer(inplace=True)
        self.conv2 = nn.Conv2d(first_planes, width, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(width)
        self.act2 = act_layer(inplace=True)
        self.conv3 = nn.Conv2d(width, outplanes, kernel_size=1, bias=False)
        self.bn3 = norm_layer(outplanes)
        self.act3 = act_layer(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def zero_init_last(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        r = engine.current_region()
        late = SHORTCUT_LATE and self.downsample is not None and not SHORTCUT_BRANCH
        shortcut = None if late else _shortcut_branch(r, x, self.downsample)
        y = EF.conv_bn_act(r, x, self.conv1, self.bn1, relu=True)
        y = EF.conv_bn_act(r, y, self.conv2, self.bn2, relu=True)
        if late:
            shortcut = _shortcut_branch(r, x, self.downsample)
        return EF.conv_bn_act(r, y, self.conv3, self.bn3, relu=True, shortcut=shortcut)


# measured on ResNet-50 (B=256): no gain (23.8 vs 23.7 ms/step) — the main-stream kernels already fill the GPU; the
# mechanism pays off where the main chain is under-filled (HRNet's parallel branches)
import os as _os
SHORTCUT_BRANCH = _os.environ.get('TOK_SHORTCUT_BRANCH', '0') == '1'
# The projection recorded LAST before the unit that adds it: in backward its (strided) data gradient then opens the
# gradient of the block input and conv1's dense 1x1 data gradient closes it — the accumulate + ReLU-mask pass over the whole
# tensor runs in the ring kernel's staged epilogue instead of the parity-class kernel's read-modify-write of untouched pixels
SHORTCUT_LATE = _os.environ.get('TOK_SHORTCUT_LATE', '1') == '1'


def _shortcut_branch(r, x, downsample):
    """The projection shortcut does not depend on the conv1..conv3 chain: it is recorded first, on branch stream 1, and
    runs beside the chain in forward and in backward; the unit that adds it waits for it."""
    if downsample is None:
        return x
    with r.branch(1 if SHORTCUT_BRANCH else 0) as br:
        return br.publish(_run_downsample(r, x, downsample))


def _run_downsample(r, x, downsample):
    """Shortcut projection: an ``nn.Sequential(conv, bn)`` ([timm] downsample_conv, hrnet.py) or a ConvBnAct brick
    without activation (necks/classification/hrnet.py:62-69)."""
    if hasattr(downsample, 'run'):
        return downsample.run(r, x)
    if len(downsample) == 3:       # [timm] downsample_avg: (AvgPool2d | Identity, 1x1 conv, norm)
        if isinstance(downsample[0], nn.AvgPool2d):
            x = EF.avg_pool_2x2(r, x)
        return EF.conv_bn_act(r, x, downsample[1], downsample[2], relu=False)
    return EF.conv_bn_act(r, x, downsample[0], downsample[1], relu=False)


def _unsupported(**flags):
    bad = [k for k, v in flags.items() if v]
    if bad:
        raise NotImplementedError(f'torchok_amd ResNet: {bad} not built (plain v1.5 ResNets only)')


def downsample_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None,
                    norm_layer=None):
    norm_layer = norm_layer or nn.BatchNorm2d
    kernel_size = 1 if stride == 1 and dilation == 1 else kernel_size
    first_dilation = (first_dilation or dilation) if kernel_size > 1 else 1
    p = get_padding(kernel_size, stride, first_dilation)
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=p, dilation=first_dilation,
                  bias=False),
        norm_layer(out_channels))


def downsample_avg(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None, norm_layer=None):
    """[timm] downsample_avg: 2x2 average pool (only where the block strides) -> 1x1 stride-1 conv -> norm."""
    norm_layer = norm_layer or nn.BatchNorm2d
    pool = nn.Identity()
    if stride != 1:
        if stride != 2 or dilation != 1:
            raise NotImplementedError('torchok_amd ResNet: avg_down with stride 2, no dilation')
        pool = nn.AvgPool2d(2, stride, ceil_mode=True, count_include_pad=False)
    return nn.Sequential(pool, nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0, bias=False),
                         norm_layer(out_channels))


def make_blocks(block_fn, channels, block_repeats, inplanes, reduce_first=1, output_stride=32,
                down_kernel_size=1, avg_down=False, drop_block_rate=0., drop_path_rate=0., **kwargs):
    _unsupported(drop_block_rate=drop_block_rate, drop_path_rate=drop_path_rate, output_stride=output_stride != 32)
    no_downsample_stages = kwargs.pop('no_downsample_stages', [0])
    stages, feature_info = [], []
    net_stride = 4
    for stage_idx, (planes, num_blocks) in enumerate(zip(channels, block_repeats)):
        stage_name = f'layer{stage_idx + 1}'
        stride = 1 if stage_idx in no_downsample_stages else 2
        net_stride *= stride
        downsample = None
        if stride != 1 or inplanes != planes * block_fn.expansion:
            make = downsample_avg if avg_down else downsample_conv
            downsample = make(inplanes, planes * block_fn.expansion, kernel_size=down_kernel_size,
                              stride=stride, norm_layer=kwargs.get('norm_layer'))
        blocks = []
        for block_idx in range(num_blocks):
            blocks.append(block_fn(inplanes, planes, stride if block_idx == 0 else 1,
                                   downsample if block_idx == 0 else None, reduce_first=reduce_first, **kwargs))
            inplanes = planes * block_fn.expansion
        stages.append((stage_name, nn.Sequential(*blocks)))
        feature_info.append(dict(num_chs=inplanes, reduction=net_stride, module=stage_name))
    return stages, feature_info


class ResNet(BaseBackbone):
    def __init__(self, block, layers, in_channels=3, output_stride=32, cardinality=1, base_width=64,
                 stem_width=64, stem_type='', replace_stem_pool=False, block_reduce_first=1, down_kernel_size=1,
                 avg_down=False, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d, aa_layer=None, drop_path_rate=0.,
                 drop_block_rate=0., zero_init_last=True, block_args=None):
        super().__init__(in_channels=in_channels)
        block_args = block_args or dict()
        if output_stride not in (8, 16, 32):
            raise ValueError('`output_stride` must be in (8, 16, 32)')
        _unsupported(stem_type=stem_type not in ('', 'deep', 'deep_tiered'), replace_stem_pool=replace_stem_pool,
                     aa_layer=aa_layer, act_layer=act_layer is not nn.ReLU, norm_layer=norm_layer is not nn.BatchNorm2d)
        deep_stem = 'deep' in stem_type
        inplanes = stem_width * 2 if deep_stem else 64
        if deep_stem:      # three 3x3 convs (resnet.py:475-486): conv1 becomes a Sequential, bn1 / act1 follow its last conv
            stem_chs = (3 * (stem_width // 4), stem_width) if 'tiered' in stem_type else (stem_width, stem_width)
            self.conv1 = nn.Sequential(
                nn.Conv2d(in_channels, stem_chs[0], 3, stride=2, padding=1, bias=False), norm_layer(stem_chs[0]),
                act_layer(inplace=True),
                nn.Conv2d(stem_chs[0], stem_chs[1], 3, stride=1, padding=1, bias=False), norm_layer(stem_chs[1]),
                act_layer(inplace=True),
                nn.Conv2d(stem_chs[1], inplanes, 3, stride=1, padding=1, bias=False))
        else:
            self.conv1 = nn.Conv2d(in_channels, inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(inplanes)
        self.act1 = act_layer(inplace=True)
        self.feature_info = [dict(num_chs=inplanes, reduction=2, module='act1')]
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)

        channels = [64, 128, 256, 512]
        stage_modules, stage_feature_info = make_blocks(
            block, channels, layers, inplanes, cardinality=cardinality, base_width=base_width,
            output_stride=output_stride, reduce_first=block_reduce_first, avg_down=avg_down,
            down_kernel_size=down_kernel_size, act_layer=act_layer, norm_layer=norm_layer, aa_layer=aa_layer,
            drop_block_rate=drop_block_rate, drop_path_rate=drop_path_rate, **block_args)
        for stage in stage_modules:
            self.add_module(*stage)
        self.feature_info.extend(stage_feature_info)
        self._out_channels = 512 * block.expansion
        self.create_hooks()
        self.init_weights(zero_init_last=zero_init_last)
        # masters live in [k][r][s][c] order (what the MFMA packs and the wgrad kernel use);
        # logical shapes / state_dict are unchanged
        self.to(memory_format=torch.channels_last)

    def init_weights(self, zero_init_last=True):
        for n, m in self.named_modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_last:
            for m in self.modules():
                if hasattr(m, 'zero_init_last'):
                    m.zero_init_last()

    def _run(self, r, x: torch.Tensor, all_features: bool = True):
        t = r.input(x, c_pad_to=4 if x.shape[1] <= 4 else 8)
        feats = []
        last = self.conv1
        if isinstance(self.conv1, nn.Sequential):            # deep stem: two conv-bn-relu units in front of the last conv
            t = EF.conv_bn_act(r, t, self.conv1[0], self.conv1[1], relu=True)
            t = EF.conv_bn_act(r, t, self.conv1[3], self.conv1[4], relu=True)
            last = self.conv1[6]
        if all_features or not EF.FUSE_STEM_POOL:
            t = EF.conv_bn_act(r, t, last, self.bn1, relu=True)
            feats.append(t)                      # 'act1', the first entry of feature_info
            t = EF.max_pool_3x3_s2(r, t)
        else:
            # nobody asked for act1: bn1 + ReLU + max-pool in one pass, the 112 x 112 activated map is never stored
            t = EF.conv_bn_act(r, t, last, self.bn1, relu=True, pool=True)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            t = layer(t)
            feats.append(t)
        return feats

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        with engine.region() as r:
            feats = self._run(r, x, all_features=False)
            return r.output(feats[-1])

    def forward_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        with engine.region() as r:
            feats = self._run(r, x)
            outs = r.output(*feats)
        return [x] + list(outs)

    def get_stages(self, stage: int) -> nn.Module:
        output = [self.conv1, self.bn1, self.act1, self.maxpool]
        layers = [self.layer1, self.layer2, self.layer3, self.layer4]
        return nn.ModuleList(output + layers[:stage])


def _create_resnet(variant, pretrained=False, **kwargs):
    # [timm] build_model_with_cfg: drops kwargs_filter keys, instantiates, loads weights iff pretrained
    for k in ('num_classes', 'global_pool', 'in_chans'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a download (no network here); pass '
                           f'pretrained=false and use task.load_checkpoint for local checkpoints')
    return ResNet(**kwargs)


@BACKBONES.register_class
def resnet18(pretrained=False, **kwargs):
    return _create_resnet('resnet18', pretrained, **dict(block=BasicBlock, layers=[2, 2, 2, 2], **kwargs))


@BACKBONES.register_class
def resnet34(pretrained=False, **kwargs):
    return _create_resnet('resnet34', pretrained, **dict(block=BasicBlock, layers=[3, 4, 6, 3], **kwargs))


@BACKBONES.register_class
def resnet50(pretrained=False, **kwargs):
    return _create_resnet('resnet50', pretrained, **dict(block=Bottleneck, layers=[3, 4, 6, 3], **kwargs))


@BACKBONES.register_class
def resnet101(pretrained=False, **kwargs):
    return _create_resnet('resnet101', pretrained, **dict(block=Bottleneck, layers=[3, 4, 23, 3], **kwargs))


@BACKBONES.register_class
def resnet152(pretrained=False, **kwargs):
    return _create_resnet('resnet152', pretrained, **dict(block=Bottleneck, layers=[3, 8, 36, 3], **kwargs))


# ---- further plain v1.5 entrypoints of the reference (same blocks, other depths / bottleneck widths; the tv_ / ssl_ /
# swsl_ names differ from their plain twins only in the pretrained weights they would download) ----------------------
def _plain(name, block, layers, doc, default_pretrained=False, **fixed):
    def entry(pretrained=default_pretrained, **kwargs):
        return _create_resnet(name, pretrained, **dict(block=block, layers=layers, **fixed, **kwargs))
    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = doc
    return BACKBONES.register_class(entry)


resnet26 = _plain('resnet26', Bottleneck, [2, 2, 2, 2], 'resnet.py:623-628')
resnet200 = _plain('resnet200', Bottleneck, [3, 24, 36, 3], 'resnet.py:707-712')
tv_resnet34 = _plain('tv_resnet34', BasicBlock, [3, 4, 6, 3], 'resnet.py:724-729')
tv_resnet50 = _plain('tv_resnet50', Bottleneck, [3, 4, 6, 3], 'resnet.py:732-737')
tv_resnet101 = _plain('tv_resnet101', Bottleneck, [3, 4, 23, 3], 'resnet.py:740-745')
tv_resnet152 = _plain('tv_resnet152', Bottleneck, [3, 8, 36, 3], 'resnet.py:748-753')
wide_resnet50_2 = _plain('wide_resnet50_2', Bottleneck, [3, 4, 6, 3], 'resnet.py:756-765 (bottleneck width x 2)', base_width=128)
wide_resnet101_2 = _plain('wide_resnet101_2', Bottleneck, [3, 4, 23, 3], 'resnet.py:768-776', base_width=128)
ssl_resnet18 = _plain('ssl_resnet18', BasicBlock, [2, 2, 2, 2], 'resnet.py:881-888', default_pretrained=True)
ssl_resnet50 = _plain('ssl_resnet50', Bottleneck, [3, 4, 6, 3], 'resnet.py:891-898', default_pretrained=True)
swsl_resnet18 = _plain('swsl_resnet18', BasicBlock, [2, 2, 2, 2], 'resnet.py:942-949', default_pretrained=True)
swsl_resnet50 = _plain('swsl_resnet50', Bottleneck, [3, 4, 6, 3], 'resnet.py:953-960', default_pretrained=True)
# 'd' = deep stem (three 3x3 convs, width 32) + average-pool shortcut projections; 't' = tiered deep stem (24, 32)
_D = dict(stem_width=32, stem_type='deep', avg_down=True)
_T = dict(stem_width=32, stem_type='deep_tiered', avg_down=True)
resnet18d = _plain('resnet18d', BasicBlock, [2, 2, 2, 2], 'resnet.py:597-603', **_D)
resnet34d = _plain('resnet34d', BasicBlock, [3, 4, 6, 3], 'resnet.py:614-620', **_D)
resnet26d = _plain('resnet26d', Bottleneck, [2, 2, 2, 2], 'resnet.py:640-645', **_D)
resnet26t = _plain('resnet26t', Bottleneck, [2, 2, 2, 2], 'resnet.py:631-637', **_T)
resnet50d = _plain('resnet50d', Bottleneck, [3, 4, 6, 3], 'resnet.py:656-662', **_D)
resnet50t = _plain('resnet50t', Bottleneck, [3, 4, 6, 3], 'resnet.py:665-671', **_T)
resnet101d = _plain('resnet101d', Bottleneck, [3, 4, 23, 3], 'resnet.py:682-687', **_D)
resnet152d = _plain('resnet152d', Bottleneck, [3, 8, 36, 3], 'resnet.py:698-704', **_D)
resnet200d = _plain('resnet200d', Bottleneck, [3, 24, 36, 3], 'resnet.py:715-721', **_D)
