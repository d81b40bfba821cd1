"""CPU restatement of the TorchOk hot path for ResNet classification:
wiring `torchok/models/backbones/resnet.py:363-551`, `base_backbone.py:14-34`,
`poolings/classification/pooling.py:7-12`, `heads/representation/linear_head.py:10-36`,
`heads/classification/classification_head.py:9-40`, `losses/base.py:7-113`,
`tasks/classification.py:75-119`, `tasks/base.py:125-133`; optimizers = torch.optim (what the
reference registers, `optim/optimizers/__init__.py:9-19`).  fp32, plain PyTorch ops.
TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import timm_min as T


def make_blocks(block_fn, channels, block_repeats, inplanes, avg_down=False, **kwargs):
    # resnet.py:363-405 with reduce_first=1, output_stride=32, down_kernel_size=1, no drop
    stages, feature_info = [], []
    net_stride = 4
    for stage_idx, (planes, num_blocks) in enumerate(zip(channels, block_repeats)):
        stride = 1 if stage_idx == 0 else 2
        net_stride *= stride
        downsample = None
        if stride != 1 or inplanes != planes * block_fn.expansion:
            down = T.downsample_avg if avg_down else T.downsample_conv                                # :383-387
            downsample = down(inplanes, planes * block_fn.expansion, kernel_size=1, stride=stride)
        blocks = []
        for block_idx in range(num_blocks):
            blocks.append(block_fn(inplanes, planes, stride if block_idx == 0 else 1,
                                   downsample if block_idx == 0 else None, **kwargs))
            inplanes = planes * block_fn.expansion
        stages.append((f'layer{stage_idx + 1}', nn.Sequential(*blocks)))
        feature_info.append(dict(num_chs=inplanes, reduction=net_stride, module=f'layer{stage_idx + 1}'))
    return stages, feature_info


class ResNet(nn.Module):
    def __init__(self, block, layers, in_channels=3, zero_init_last=True, base_width=64, stem_width=64, stem_type='',
                 avg_down=False):
        super().__init__()
        if 'deep' in stem_type:                                                                  # :472-486
            chs = (3 * (stem_width // 4), stem_width) if 'tiered' in stem_type else (stem_width, stem_width)
            self.conv1 = nn.Sequential(
                nn.Conv2d(in_channels, chs[0], 3, stride=2, padding=1, bias=False), nn.BatchNorm2d(chs[0]), nn.ReLU(inplace=True),
                nn.Conv2d(chs[0], chs[1], 3, stride=1, padding=1, bias=False), nn.BatchNorm2d(chs[1]), nn.ReLU(inplace=True),
                nn.Conv2d(chs[1], stem_width * 2, 3, stride=1, padding=1, bias=False))
            assert stem_width * 2 == 64
        else:
            self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)  # :488
        self.bn1 = nn.BatchNorm2d(64)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)                          # :510
        stages, _ = make_blocks(block, [64, 128, 256, 512], layers, 64, avg_down=avg_down, base_width=base_width)     # :515-518
        for s in stages:
            self.add_module(*s)
        self.out_channels = 512 * block.expansion
        for m in self.modules():                                                                 # :529-539
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_last:
            for m in self.modules():
                if hasattr(m, 'zero_init_last'):
                    m.zero_init_last()

    def forward_features(self, x):
        feats = [x]
        x = self.act1(self.bn1(self.conv1(x)))
        feats.append(x)
        x = self.maxpool(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            feats.append(x)
        return feats

    def forward(self, x):                                                                        # :541-551
        return self.forward_features(x)[-1]


def resnet18(**kw):
    return ResNet(T.BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(**kw):
    return ResNet(T.BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(**kw):
    return ResNet(T.Bottleneck, [3, 4, 6, 3], **kw)


def resnet26(**kw):                                   # resnet.py:623-628
    return ResNet(T.Bottleneck, [2, 2, 2, 2], **kw)


def wide_resnet50_2(**kw):                            # resnet.py:756-765: bottleneck width x2, outer 1x1 widths unchanged
    return ResNet(T.Bottleneck, [3, 4, 6, 3], base_width=128, **kw)


def resnet18d(**kw):                                  # resnet.py:597-603
    return ResNet(T.BasicBlock, [2, 2, 2, 2], stem_width=32, stem_type='deep', avg_down=True, **kw)


def resnet26t(**kw):                                  # resnet.py:631-637
    return ResNet(T.Bottleneck, [2, 2, 2, 2], stem_width=32, stem_type='deep_tiered', avg_down=True, **kw)


def resnet50d(**kw):                                  # resnet.py:656-662
    return ResNet(T.Bottleneck, [3, 4, 6, 3], stem_width=32, stem_type='deep', avg_down=True, **kw)


BACKBONES = dict(resnet18d=resnet18d, resnet26t=resnet26t, resnet50d=resnet50d, resnet18=resnet18, resnet34=resnet34, resnet50=resnet50, resnet26=resnet26,
                 wide_resnet50_2=wide_resnet50_2, tv_resnet34=resnet34, tv_resnet50=resnet50, ssl_resnet18=resnet18,
                 swsl_resnet50=resnet50)


class ClassificationModel(nn.Module):
    """backbone -> Pooling('avg') -> ClassificationHead; same child names as ClassificationTask
    so state_dicts are interchangeable with the build (`backbone.*`, `head.fc.*`)."""

    def __init__(self, backbone: str, num_classes: int, **backbone_kw):
        super().__init__()
        self.backbone = BACKBONES[backbone](**backbone_kw)
        self.pooling = T.SelectAdaptivePool2d(1, 'avg', flatten=True)
        self.head = nn.Module()
        self.head.fc = nn.Linear(self.backbone.out_channels, num_classes)
        self.num_classes = num_classes

    def forward_with_gt(self, batch):
        features = self.backbone(batch['image'])
        embeddings = self.pooling(features)
        prediction = self.head.fc(embeddings)
        if self.num_classes == 1:
            prediction = prediction[..., 0]
        return {'embeddings': embeddings, 'prediction': prediction, 'target': batch['target']}


def joint_loss(losses, mappings, tags, weights, normalize_weights, **outputs):
    """losses/base.py:43-85."""
    n_spec = len([w for w in weights if w is not None])
    if n_spec > 0 and n_spec != len(losses):
        raise ValueError('Loss weights must be either specified for each loss function or not specified for any loss function')
    w = [1.] * len(losses) if n_spec == 0 else list(weights)
    if normalize_weights:
        w = [x / sum(w) for x in w]
    total, tagged = 0., {}
    for fn, mp, tag, wi in zip(losses, mappings, tags, w):
        val = fn(**{k: outputs[v] for k, v in mp.items()})
        total = total + val * wi
        if tag is not None:
            tagged[tag] = val
    return total, tagged


def training_step(model: ClassificationModel, batch, optimizer=None):
    """tasks/base.py:125-133 followed by Lightning's backward + optimizer.step()."""
    out = model.forward_with_gt(batch)
    total, _ = joint_loss([nn.CrossEntropyLoss()], [dict(input='prediction', target='target')], [None], [None],
                          True, **out)
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    total.backward()
    if optimizer is not None:
        optimizer.step()
    return total.detach(), out
