"""CPU restatement of the HRNet segmentation rows, fp32, plain PyTorch.
  * [timm 0.6.13] `timm.models.hrnet`: `cfg_cls`, `HighResolutionModule`, `_BN_MOMENTUM`, `blocks_dict`
    (absent offline; semantics per SURVEY.md App. A.2) — imported by `torchok/models/backbones/hrnet.py:13`
  * wiring of `torchok/models/backbones/hrnet.py:52-260` (HighResolutionNet),
    `necks/segmentation/hrnet.py:16-43`, `heads/segmentation/base.py:12-42`, `tasks/segmentation.py:60-93`
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/hrnet_seg_step.npz (tests/golden/gen_golden.py runs the reference's
own hrnet.py / neck / head on the stubbed timm and asserts this file reproduces it bit for bit)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .timm_min import BasicBlock, Bottleneck

_BN_MOMENTUM = 0.1
blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


def _stage(modules, block, blocks, channels):
    return dict(NUM_MODULES=modules, NUM_BRANCHES=len(channels), BLOCK=block, NUM_BLOCKS=tuple(blocks),
                NUM_CHANNELS=tuple(channels), FUSE_METHOD='SUM')


def _cfg(stage1, stage2, stage3, stage4):
    return dict(STEM_WIDTH=64, STAGE1=_stage(1, 'BOTTLENECK', *stage1), STAGE2=_stage(stage2[0], 'BASIC', *stage2[1:]),
                STAGE3=_stage(stage3[0], 'BASIC', *stage3[1:]), STAGE4=_stage(stage4[0], 'BASIC', *stage4[1:]))


cfg_cls = {
    'hrnet_w18_small': _cfg(((1,), (32,)), (1, (2, 2), (16, 32)), (1, (2, 2, 2), (16, 32, 64)),
                            (1, (2, 2, 2, 2), (16, 32, 64, 128))),
    'hrnet_w18_small_v2': _cfg(((2,), (64,)), (1, (2, 2), (18, 36)), (3, (2, 2, 2), (18, 36, 72)),
                               (2, (2, 2, 2, 2), (18, 36, 72, 144))),
}
for _w in (18, 30, 32, 40, 44, 48, 64):
    cfg_cls[f'hrnet_w{_w}'] = _cfg(((4,), (64,)), (1, (4, 4), (_w, 2 * _w)), (4, (4, 4, 4), (_w, 2 * _w, 4 * _w)),
                                   (3, (4, 4, 4, 4), (_w, 2 * _w, 4 * _w, 8 * _w)))


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, blocks, num_blocks, num_in_chs, num_channels, fuse_method, multi_scale_output=True):
        super().__init__()
        assert num_branches == len(num_blocks) == len(num_channels) == len(num_in_chs)
        self.num_in_chs = num_in_chs
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList(self._branch(i, blocks, num_blocks, num_channels) for i in range(num_branches))
        self.fuse_layers = self._fuse_layers()
        self.fuse_act = nn.ReLU(False)

    def _branch(self, i, block, num_blocks, num_channels):
        width = num_channels[i] * block.expansion
        down = None
        if self.num_in_chs[i] != width:
            down = nn.Sequential(nn.Conv2d(self.num_in_chs[i], width, kernel_size=1, stride=1, bias=False),
                                 nn.BatchNorm2d(width, momentum=_BN_MOMENTUM))
        seq = [block(self.num_in_chs[i], num_channels[i], 1, down)]
        self.num_in_chs[i] = width
        seq += [block(width, num_channels[i]) for _ in range(num_blocks[i] - 1)]
        return nn.Sequential(*seq)

    def _fuse_layers(self):
        if self.num_branches == 1:
            return nn.Identity()
        ch = self.num_in_chs
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, 1, 0, bias=False),
                                             nn.BatchNorm2d(ch[i], momentum=_BN_MOMENTUM),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(nn.Identity())
                else:
                    steps = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            steps.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 3, 2, 1, bias=False),
                                                       nn.BatchNorm2d(ch[i], momentum=_BN_MOMENTUM)))
                        else:
                            steps.append(nn.Sequential(nn.Conv2d(ch[j], ch[j], 3, 2, 1, bias=False),
                                                       nn.BatchNorm2d(ch[j], momentum=_BN_MOMENTUM), nn.ReLU(False)))
                    row.append(nn.Sequential(*steps))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_in_chs(self):
        return self.num_in_chs

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        for i, branch in enumerate(self.branches):
            x[i] = branch(x[i])
        out = []
        for i, row in enumerate(self.fuse_layers):
            y = x[0] if i == 0 else row[0](x[0])
            for j in range(1, self.num_branches):
                y = y + (x[j] if i == j else row[j](x[j]))
            out.append(self.fuse_act(y))
        return out


class HRNet(nn.Module):
    """`HighResolutionNet` of torchok/models/backbones/hrnet.py (same child names)."""

    def __init__(self, variant, in_channels=3):
        super().__init__()
        cfg = cfg_cls[variant]
        self.out_encoder_channels = cfg['STAGE4']['NUM_CHANNELS']
        sw = cfg['STEM_WIDTH']
        self.conv1 = nn.Conv2d(in_channels, sw, kernel_size=3, stride=2, padding=1, bias=False)     # :63-69
        self.bn1 = nn.BatchNorm2d(sw, momentum=_BN_MOMENTUM)
        self.conv2 = nn.Conv2d(sw, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=_BN_MOMENTUM)
        s1 = cfg['STAGE1']
        block = blocks_dict[s1['BLOCK']]
        self.layer1 = self._layer(block, 64, s1['NUM_CHANNELS'][0], s1['NUM_BLOCKS'][0])            # :71-76
        pre = [block.expansion * s1['NUM_CHANNELS'][0]]
        for n in (2, 3, 4):                                                                           # :78-100
            sc = cfg[f'STAGE{n}']
            block = blocks_dict[sc['BLOCK']]
            ch = [c * block.expansion for c in sc['NUM_CHANNELS']]
            setattr(self, f'transition{n - 1}', self._transition(pre, ch))
            mods = []
            for _ in range(sc['NUM_MODULES']):
                mods.append(HighResolutionModule(sc['NUM_BRANCHES'], block, sc['NUM_BLOCKS'], ch, sc['NUM_CHANNELS'],
                                                 sc['FUSE_METHOD'], True))
                ch = mods[-1].get_num_in_chs()
            setattr(self, f'stage{n}', nn.Sequential(*mods))
            pre = ch
        for m in self.modules():                                                                      # :104-112
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _cbr(cin, cout, stride):
        return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=False), nn.BatchNorm2d(cout, momentum=_BN_MOMENTUM),
                             nn.ReLU(inplace=True))

    @classmethod
    def _transition(cls, pre, cur):                                                                   # :114-140
        out = []
        for i, c in enumerate(cur):
            if i < len(pre):
                out.append(cls._cbr(pre[i], c, 1) if c != pre[i] else nn.Identity())
            else:
                n = i + 1 - len(pre)
                out.append(nn.Sequential(*[cls._cbr(pre[-1], c if j == n - 1 else pre[-1], 2) for j in range(n)]))
        return nn.ModuleList(out)

    @staticmethod
    def _layer(block, cin, cout, n):                                                                  # :142-166
        down = None
        if cin != cout * block.expansion:
            down = nn.Sequential(nn.Conv2d(cin, cout * block.expansion, kernel_size=1, stride=1, bias=False),
                                 nn.BatchNorm2d(cout * block.expansion, momentum=_BN_MOMENTUM))
        seq = [block(cin, cout, 1, down)] + [block(cout * block.expansion, cout) for _ in range(n - 1)]
        return nn.Sequential(*seq)

    def forward(self, x):                                                                              # :197-233
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        yl = self.stage2([t(x) for t in self.transition1])
        for tr, st in ((self.transition2, self.stage3), (self.transition3, self.stage4)):
            yl = st([yl[i] if isinstance(t, nn.Identity) else t(yl[-1]) for i, t in enumerate(tr)])
        return yl

    def forward_features(self, x):                                                                     # :235-241
        return [x] + self.forward(x)


class SegmentationModel(nn.Module):
    """backbone -> HRNetSegmentationNeck -> SegmentationHead with the child names of SegmentationTask."""

    def __init__(self, variant, num_classes):
        super().__init__()
        self.backbone = HRNet(variant)
        c = sum(self.backbone.out_encoder_channels)
        self.neck = nn.Module()
        self.neck.convbnact = nn.Module()
        self.neck.convbnact.conv = nn.Conv2d(c, c, kernel_size=1, bias=False)
        self.neck.convbnact.bn = nn.BatchNorm2d(c)
        self.head = nn.Module()
        self.head.classifier = nn.Conv2d(c, num_classes, kernel_size=1)
        self.num_classes = num_classes

    def neck_forward(self, feats):                       # necks/segmentation/hrnet.py:31-43
        image, x0, x1, x2, x3 = feats
        size = (x0.size(2), x0.size(3))
        ups = [F.interpolate(t, size=size, mode='bilinear', align_corners=False) for t in (x1, x2, x3)]
        f = torch.cat([x0] + ups, 1)
        return [image, F.relu(self.neck.convbnact.bn(self.neck.convbnact.conv(f)))]

    def head_forward(self, x):                           # heads/segmentation/base.py:31-42
        image, f = x
        logits = F.interpolate(self.head.classifier(f), size=image.shape[2:], mode='bilinear')
        return logits[:, 0] if self.num_classes == 1 else logits

    def forward_with_gt(self, batch):                    # tasks/segmentation.py:60-93
        feats = self.backbone.forward_features(batch['image'])
        return {'prediction': self.head_forward(self.neck_forward(feats)), 'target': batch['target']}


class ClassificationNeck(nn.Module):
    """`HRNetClassificationNeck` (necks/classification/hrnet.py:12-92), including its forward as written: the loop
    OVERWRITES y with incre_modules[i + 1](x[i + 1]) (:88-90), so only the last branch reaches final_layer."""

    class _CBA(nn.Module):          # models/modules/bricks/convbnact.py
        def __init__(self, cin, cout, k, pad, stride, act=True):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False)
            self.bn = nn.BatchNorm2d(cout)
            self.act = nn.ReLU(inplace=True) if act else nn.Identity()

        def forward(self, x):
            return self.act(self.bn(self.conv(x)))

    def __init__(self, in_channels):
        super().__init__()
        hc, e = [32, 64, 128, 256], Bottleneck.expansion
        self.incre_modules = nn.ModuleList(
            nn.Sequential(Bottleneck(c, hc[i], 1, self._CBA(c, hc[i] * e, 1, 0, 1, act=False) if c != hc[i] * e else None))
            for i, c in enumerate(in_channels))
        self.downsamp_modules = nn.ModuleList(self._CBA(hc[i] * e, hc[i + 1] * e, 3, 1, 2) for i in range(len(in_channels) - 1))
        self.final_layer = self._CBA(hc[3] * e, 2048, 1, 0, 1)

    def forward(self, x):
        y = self.incre_modules[0](x[0])
        for i in range(len(self.downsamp_modules)):
            y = self.downsamp_modules[i](y)
            if i + 1 < len(x):
                y = self.incre_modules[i + 1](x[i + 1])
        return self.final_layer(y)
