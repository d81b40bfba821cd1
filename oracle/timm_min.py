"""Restatement of the [timm 0.6.13] pieces the reference's ResNet hot path imports
(`torchok/models/backbones/resnet.py:12-14`: BasicBlock, Bottleneck, downsample_conv, drop_blocks,
create_aa, DropPath; `poolings/classification/pooling.py:1`: SelectAdaptivePool2d).
Plain PyTorch, CPU, fp32.  Semantics per SURVEY.md Appendix A.1 / A.4 (timm itself is not
available offline).  TEST INFRASTRUCTURE ONLY."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def get_padding(kernel_size, stride, dilation=1):
    return ((stride - 1) + dilation * (kernel_size - 1)) // 2


def create_aa(aa_layer, channels=None, stride=2, enable=True):
    if not aa_layer or not enable:
        return nn.Identity()
    raise NotImplementedError


def drop_blocks(drop_prob=0.):
    return [None, None, None, None]


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


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64,
                 reduce_first=1, dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d,
                 attn_layer=None, aa_layer=None, drop_block=None, drop_path=None):
        super().__init__()
        assert cardinality == 1 and base_width == 64
        first_planes = planes // reduce_first
        outplanes = planes * self.expansion
        first_dilation = first_dilation or dilation
        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=3, stride=stride, padding=first_dilation,
                               dilation=first_dilation, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = act_layer(inplace=True)
        self.conv2 = nn.Conv2d(first_planes, outplanes, kernel_size=3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(outplanes)
        self.act2 = act_layer(inplace=True)
        self.downsample = downsample
        self.drop_path = drop_path

    def zero_init_last(self):
        nn.init.zeros_(self.bn2.weight)

    def forward(self, x):
        shortcut = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        if self.drop_path is not None:
            x = self.drop_path(x)
        if self.downsample is not None:
            shortcut = self.downsample(shortcut)
        x = x + shortcut
        return self.act2(x)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, cardinality=1, base_width=64,
                 reduce_first=1, dilation=1, first_dilation=None, act_layer=nn.ReLU, norm_layer=nn.BatchNorm2d,
                 attn_layer=None, aa_layer=None, drop_block=None, drop_path=None):
        super().__init__()
        width = int(math.floor(planes * (base_width / 64)) * cardinality)
        first_planes = width // reduce_first
        outplanes = planes * self.expansion
        first_dilation = first_dilation or dilation
        self.conv1 = nn.Conv2d(inplanes, first_planes, kernel_size=1, bias=False)
        self.bn1 = norm_layer(first_planes)
        self.act1 = act_layer(inplace=True)
        self.conv2 = nn.Conv2d(first_planes, width, kernel_size=3, stride=stride, padding=first_dilation,
                               dilation=first_dilation, groups=cardinality, bias=False)
        self.bn2 = norm_layer(width)
        self.act2 = act_layer(inplace=True)
        self.conv3 = nn.Conv2d(width, outplanes, kernel_size=1, bias=False)
        self.bn3 = norm_layer(outplanes)
        self.act3 = act_layer(inplace=True)
        self.downsample = downsample
        self.drop_path = drop_path

    def zero_init_last(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        shortcut = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.act2(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        if self.drop_path is not None:
            x = self.drop_path(x)
        if self.downsample is not None:
            shortcut = self.downsample(shortcut)
        x = x + shortcut
        return self.act3(x)


def downsample_conv(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None,
                    norm_layer=None):
    norm_layer = norm_layer or nn.BatchNorm2d
    kernel_size = 1 if stride == 1 and dilation == 1 else kernel_size
    first_dilation = (first_dilation or dilation) if kernel_size > 1 else 1
    p = get_padding(kernel_size, stride, first_dilation)
    return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=p,
                                   dilation=first_dilation, bias=False), norm_layer(out_channels))


def downsample_avg(in_channels, out_channels, kernel_size, stride=1, dilation=1, first_dilation=None, norm_layer=None):
    """[timm 0.6.13] resnet.downsample_avg: AvgPool2d(2, stride, ceil_mode=True,
    count_include_pad=False) where the block strides (Identity at stride 1), then a 1x1 stride-1 conv and the norm."""
    norm_layer = norm_layer or nn.BatchNorm2d
    avg_stride = stride if dilation == 1 else 1
    if stride == 1 and dilation == 1:
        pool = nn.Identity()
    else:
        assert avg_stride != 1 or dilation == 1, 'AvgPool2dSame (dilated avg_down) is not restated'
        pool = nn.AvgPool2d(2, avg_stride, ceil_mode=True, count_include_pad=False)
    return nn.Sequential(pool, nn.Conv2d(in_channels, out_channels, 1, stride=1, padding=0, bias=False),
                         norm_layer(out_channels))


class _AvgMax(nn.Module):
    """[timm 0.6.13] layers/adaptive_avgmax_pool.py: adaptive_avgmax_pool2d / adaptive_catavgmax_pool2d."""

    def __init__(self, output_size, cat):
        super().__init__()
        self.output_size, self.cat = output_size, cat

    def forward(self, x):
        x_avg = nn.functional.adaptive_avg_pool2d(x, self.output_size)
        x_max = nn.functional.adaptive_max_pool2d(x, self.output_size)
        return torch.cat((x_avg, x_max), 1) if self.cat else 0.5 * (x_avg + x_max)


class SelectAdaptivePool2d(nn.Module):
    """[timm 0.6.13] SelectAdaptivePool2d: '' identity, 'avg', 'max', 'avgmax' = 0.5 * (avg + max),
    'catavgmax' = cat(avg, max, dim=1); then Flatten(1) when flatten=True."""

    def __init__(self, output_size=1, pool_type='fast', flatten=False):
        super().__init__()
        self.pool_type = pool_type or ''
        if pool_type == '':
            self.pool = nn.Identity()
        elif pool_type == 'avg':
            self.pool = nn.AdaptiveAvgPool2d(output_size)
        elif pool_type == 'max':
            self.pool = nn.AdaptiveMaxPool2d(output_size)
        elif pool_type == 'avgmax':
            self.pool = _AvgMax(output_size, cat=False)
        elif pool_type == 'catavgmax':
            self.pool = _AvgMax(output_size, cat=True)
        else:
            assert False, 'Invalid pool type: %s' % pool_type
        self.flatten = nn.Flatten(1) if flatten else nn.Identity()

    def feat_mult(self):
        return 2 if self.pool_type == 'catavgmax' else 1

    def forward(self, x):
        return self.flatten(self.pool(x))


class FeatureHooks:
    """[timm] timm.models.features.FeatureHooks: forward hooks stashing the outputs of the
    modules named in feature_info; get_output(device) returns and clears them."""

    def __init__(self, hooks, named_modules, out_map=None, default_hook_type='forward'):
        modules = {k: v for k, v in named_modules}
        self._store = {}
        for h in hooks:
            name = h['module']
            modules[name].register_forward_hook(lambda m, i, o, name=name: self._store.__setitem__(name, o))

    def get_output(self, device):
        from collections import OrderedDict
        out = OrderedDict(self._store)
        self._store = {}
        return out


def build_model_with_cfg(model_cls, variant, pretrained, pretrained_strict=False, kwargs_filter=(), **kwargs):
    for k in kwargs_filter or ():
        kwargs.pop(k, None)
    assert not pretrained
    if 'model_cfg' in kwargs:          # timm: model_cls(cfg=model_cfg, **kwargs)
        kwargs['cfg'] = kwargs.pop('model_cfg')
    return model_cls(**kwargs)
