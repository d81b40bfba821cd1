"""HRNet backbones on the MI355X engine.

Mirrors the reference wiring ``torchok/models/backbones/hrnet.py``: ``HighResolutionNet.__init__`` (:55-102),
``init_weights`` (:104-112), the transition / layer / stage builders (:114-195), ``forward_stages`` (:197-213),
``forward_stem`` (:226-233), ``forward_features`` (:235-241), ``get_stages`` (:243-260) and the nine
``hrnet_w*`` entrypoints (:277-328), with the [timm 0.6.13] ``HighResolutionModule`` / ``cfg_cls`` semantics
restated here.  Module / parameter names are those of timm, so reference checkpoints load.

Execution: every conv -> BN -> (ReLU) group is one engine unit (``engine.functional.conv_bn_act``); the
all-to-all fuse ``relu(sum_j fuse_ij(x_j))`` is ONE summation kernel per output branch which reads the
low-resolution BN outputs at their native size (the nearest ``nn.Upsample`` is an index shift in that kernel,
the upsampled maps never exist in HBM); the whole backbone is a single autograd node.
"""
import os
from typing import Any, Dict, List

import torch
import torch.nn as nn

from ... import engine
from ...constructor import BACKBONES
from ...engine import functional as EF
from ...engine import resample as ER
from ..base import BaseBackbone
from .resnet import BasicBlock, Bottleneck

_BRANCH0_FIRST = os.environ.get('TOK_HRNET_BRANCH0_FIRST', '0') == '1'     # A/B switch: the enqueue order of rounds 1-5
_FUSE_STREAMS = os.environ.get('TOK_HRNET_FUSE_STREAMS', '3') != '0'       # A/B switch: fuse rows on the branch streams (forward)
_FUSE_BWD_MAIN = os.environ.get('TOK_HRNET_FUSE_STREAMS', '3') != '2'      # =2: their backward on the branch streams too
_FUSE_BWD_SRC = os.environ.get('TOK_HRNET_FUSE_STREAMS', '3') == '3'       # =3 (probe): path j -> i backward on stream j, the sum's on stream i

_BN_MOMENTUM = 0.1
blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


def _stage_cfg(modules, branches, block, blocks, channels):
    return dict(NUM_MODULES=modules, NUM_BRANCHES=branches, BLOCK=block, NUM_BLOCKS=tuple(blocks),
                NUM_CHANNELS=tuple(channels), FUSE_METHOD='SUM')


def _hr_cfg(s1_blocks, s1_ch, s2, s3, s4):
    (m2, b2, c2), (m3, b3, c3), (m4, b4, c4) = s2, s3, s4
    return dict(STEM_WIDTH=64,
                STAGE1=_stage_cfg(1, 1, 'BOTTLENECK', (s1_blocks,), (s1_ch,)),
                STAGE2=_stage_cfg(m2, 2, 'BASIC', (b2,) * 2, c2),
                STAGE3=_stage_cfg(m3, 3, 'BASIC', (b3,) * 3, c3),
                STAGE4=_stage_cfg(m4, 4, 'BASIC', (b4,) * 4, c4))


def _wide(w):
    return _hr_cfg(4, 64, (1, 4, (w, 2 * w)), (4, 4, (w, 2 * w, 4 * w)), (3, 4, (w, 2 * w, 4 * w, 8 * w)))


# [timm] hrnet.cfg_cls
cfg_cls = {
    'hrnet_w18_small': _hr_cfg(1, 32, (1, 2, (16, 32)), (1, 2, (16, 32, 64)), (1, 2, (16, 32, 64, 128))),
    'hrnet_w18_small_v2': _hr_cfg(2, 64, (1, 2, (18, 36)), (3, 2, (18, 36, 72)), (2, 2, (18, 36, 72, 144))),
    'hrnet_w18': _wide(18), 'hrnet_w30': _wide(30), 'hrnet_w32': _wide(32), 'hrnet_w40': _wide(40),
    'hrnet_w44': _wide(44), 'hrnet_w48': _wide(48), 'hrnet_w64': _wide(64),
}


def _conv_bn(cin, cout, k, stride, pad, relu):
    layers = [nn.Conv2d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm2d(cout, momentum=_BN_MOMENTUM)]
    if relu:
        layers.append(nn.ReLU(inplace=False))
    return nn.Sequential(*layers)


# the last unit of a fuse path (Conv2d + BatchNorm2d, no activation) hands its RAW output + BatchNorm coefficients to the row's
# summation kernel instead of running its own apply pass (round 6: by ablation those passes cost 1.3 ms of a 64.3-ms HRNet-W48
# step; TOK_HRNET_DEFER_TERM_BN=0 restores them)
_DEFER_TERM_BN = os.environ.get('TOK_HRNET_DEFER_TERM_BN', '1') != '0'


def _run_conv_bn(r, x, seq: nn.Sequential, defer_apply: bool = False):
    """One `Conv2d, BatchNorm2d[, ReLU]` Sequential (transition / fuse building block) as one engine unit."""
    relu = len(seq) > 2 and isinstance(seq[2], nn.ReLU)
    return EF.conv_bn_act(r, x, seq[0], seq[1], relu=relu, defer_apply=defer_apply and not relu and _DEFER_TERM_BN)


class HighResolutionModule(nn.Module):
    """[timm] HighResolutionModule: parallel branches of blocks + all-to-all fuse (SUM)."""

    def __init__(self, num_branches, blocks, num_blocks, num_in_chs, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        if num_branches != len(num_blocks) or num_branches != len(num_channels) or num_branches != len(num_in_chs):
            raise ValueError('NUM_BRANCHES does not match NUM_BLOCKS / NUM_CHANNELS / NUM_INCHANNELS')
        self.num_in_chs = list(num_in_chs)
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList([self._make_one_branch(i, blocks, num_blocks, num_channels)
                                       for i in range(num_branches)])
        self.fuse_layers = self._make_fuse_layers()
        self.fuse_act = nn.ReLU(False)

    def _make_one_branch(self, i, block, num_blocks, num_channels, stride=1):
        downsample = None
        out_ch = num_channels[i] * block.expansion
        if stride != 1 or self.num_in_chs[i] != out_ch:
            downsample = nn.Sequential(nn.Conv2d(self.num_in_chs[i], out_ch, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(out_ch, momentum=_BN_MOMENTUM))
        layers = [block(self.num_in_chs[i], num_channels[i], stride, downsample)]
        self.num_in_chs[i] = out_ch
        for _ in range(1, num_blocks[i]):
            layers.append(block(self.num_in_chs[i], num_channels[i]))
        return nn.Sequential(*layers)

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return nn.Identity()
        chs = self.num_in_chs
        fuse_layers = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(chs[j], chs[i], 1, 1, 0, bias=False),
                                             nn.BatchNorm2d(chs[i], momentum=_BN_MOMENTUM),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(nn.Identity())
                else:
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        chain.append(_conv_bn(chs[j], chs[i] if last else chs[j], 3, 2, 1, relu=not last))
                    row.append(nn.Sequential(*chain))
            fuse_layers.append(nn.ModuleList(row))
        return nn.ModuleList(fuse_layers)

    def get_num_in_chs(self):
        return self.num_in_chs

    def forward(self, x: List):
        r = engine.current_region()
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        # the branches are independent until the fuse: branch 0 on the main stream, the low-resolution ones (few
        # tiles, long reductions: they cannot fill the GPU alone) each on its own branch stream beside it.
        # LOW-RESOLUTION BRANCHES FIRST (round 6): entering a branch stream records its fork event on the main stream at that
        # moment — recorded after branch 0's two dozen launches it made branches 1..3 START when branch 0 had FINISHED (per-queue
        # dump of a step: the forward of every module ran [branch 0] then [branches 1-3]; TOK_HRNET_BRANCH0_FIRST=1 restores it)
        order = range(self.num_branches) if _BRANCH0_FIRST else reversed(range(self.num_branches))
        outs = [None] * self.num_branches
        nodes = getattr(r, 'nodes', None)
        start, spans = (len(nodes) if nodes is not None else 0), {}
        for i in order:
            a = len(nodes) if nodes is not None else 0
            with r.branch(i) as br:
                outs[i] = br.publish(self.branches[i](x[i]))
            spans[i] = (a, len(nodes) if nodes is not None else 0)
        if nodes is not None and not _BRANCH0_FIRST:
            # ... but the TAPE keeps the order branch 0, 1, 2, 3 (the branches are independent blocks of it): the backward walks
            # it in reverse and starts with the low-resolution branches, as before — enqueued the other way round it was 2.5 ms
            # slower (same box: forward 23.9 vs 24.2 ms, backward 48.6 vs 46.1)
            nodes[start:] = [n for i in range(self.num_branches) for n in nodes[spans[i][0]:spans[i][1]]]
        x = outs
        # the fuse: output row i = relu(sum_j fuse_ij(x_j)) is independent of the other rows, and its result is the input of
        # branch i of the next module — so row i runs on branch stream i (round 6; it was a chain of ~24 small kernels per module
        # on the main queue, 0.6 ms forward and 1.8 ms backward with the other queues idle; TOK_HRNET_FUSE_STREAMS=0 restores that).
        # Low-resolution rows are entered first for the same reason as the branches above; the tape keeps row 0 first.
        fused = [None] * len(self.fuse_layers)
        path_spans = []
        start, spans = (len(nodes) if nodes is not None else 0), {}
        rows = range(len(self.fuse_layers))
        for i in (reversed(rows) if _FUSE_STREAMS else rows):
            row = self.fuse_layers[i]
            a = len(nodes) if nodes is not None else 0
            with r.branch(i if _FUSE_STREAMS else 0) as br:
                terms = []
                for j in range(self.num_branches):
                    if j == i:
                        terms.append((x[j], 0))
                    elif j > i:      # 1x1 conv + BN at the low resolution; the nearest upsample is folded into the sum
                        p0 = len(nodes) if nodes is not None else 0
                        terms.append((EF.conv_bn_act(r, x[j], row[j][0], row[j][1], relu=False, defer_apply=_DEFER_TERM_BN), j - i))
                        path_spans.append((p0, len(nodes) if nodes is not None else 0, j))
                    else:
                        p0 = len(nodes) if nodes is not None else 0
                        t = x[j]
                        for si, step in enumerate(row[j]):
                            t = _run_conv_bn(r, t, step, defer_apply=si == len(row[j]) - 1)
                        terms.append((t, 0))
                        path_spans.append((p0, len(nodes) if nodes is not None else 0, j))
                # output-resolution term first: it fixes the shape
                terms.sort(key=lambda ts: ts[1])
                fused[i] = br.publish(ER.fuse_sum_relu(r, terms, relu=True))
            spans[i] = (a, len(nodes) if nodes is not None else 0)
        if nodes is not None and _FUSE_STREAMS and _FUSE_BWD_SRC:
            # probe: the backward of path j -> i on the SOURCE branch's stream j (every write into d(x_j) then comes from one stream)
            for p0, p1, j in path_spans:
                for n in nodes[p0:p1]:
                    n.stream_tag = j
        if nodes is not None and _FUSE_STREAMS:
            nodes[start:] = [n for i in rows for n in nodes[spans[i][0]:spans[i][1]]]
            if _FUSE_BWD_MAIN and not _FUSE_BWD_SRC:
                # ... and the BACKWARD of the fuse rows stays on the main stream: their gradient contributions fan into the
                # branches' outputs from every row, and ordering those accumulations across four queues cost more events and
                # waits than the rows' concurrency returned
                for n in nodes[start:]:
                    n.stream_tag = 0
        return fused


class HighResolutionNet(BaseBackbone):
    def __init__(self, cfg: Dict[str, Any], in_channels: int = 3):
        super().__init__(in_channels=in_channels, out_channels=cfg['STAGE4']['NUM_CHANNELS'])
        self._out_encoder_channels = cfg['STAGE4']['NUM_CHANNELS']
        stem_width = cfg['STEM_WIDTH']
        self.conv1 = nn.Conv2d(in_channels, stem_width, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(stem_width, momentum=_BN_MOMENTUM)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(stem_width, 64, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=_BN_MOMENTUM)
        self.act2 = nn.ReLU(inplace=True)

        self.stage1_cfg = cfg['STAGE1']
        num_channels = self.stage1_cfg['NUM_CHANNELS'][0]
        block = blocks_dict[self.stage1_cfg['BLOCK']]
        self.layer1 = self._make_layer(block, 64, num_channels, self.stage1_cfg['NUM_BLOCKS'][0])
        pre = [block.expansion * num_channels]
        for idx in (2, 3, 4):
            scfg = cfg[f'STAGE{idx}']
            setattr(self, f'stage{idx}_cfg', scfg)
            block = blocks_dict[scfg['BLOCK']]
            chans = [c * block.expansion for c in scfg['NUM_CHANNELS']]
            setattr(self, f'transition{idx - 1}', self._make_transition_layer(pre, chans))
            stage, pre = self._make_stage(scfg, chans, multi_scale_output=True)
            setattr(self, f'stage{idx}', stage)
        self.init_weights()
        self.to(memory_format=torch.channels_last)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _make_transition_layer(pre: List[int], cur: List[int]) -> nn.ModuleList:
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(_relu_inplace(_conv_bn(pre[i], cur[i], 3, 1, 1, relu=True)) if cur[i] != pre[i]
                              else nn.Identity())
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    cin = pre[-1]
                    cout = cur[i] if j == i - len(pre) else cin
                    chain.append(_relu_inplace(_conv_bn(cin, cout, 3, 2, 1, relu=True)))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    @staticmethod
    def _make_layer(block, in_channels, out_channels, num_blocks, stride=1) -> nn.Sequential:
        downsample = None
        if stride != 1 or in_channels != out_channels * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(in_channels, out_channels * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(out_channels * block.expansion, momentum=_BN_MOMENTUM))
        layers = [block(in_channels, out_channels, stride, downsample)]
        in_channels = out_channels * block.expansion
        for _ in range(1, num_blocks):
            layers.append(block(in_channels, out_channels))
        return nn.Sequential(*layers)

    @staticmethod
    def _make_stage(layer_config, in_channels, multi_scale_output=True):
        block = blocks_dict[layer_config['BLOCK']]
        modules = []
        for i in range(layer_config['NUM_MODULES']):
            reset = multi_scale_output or i < layer_config['NUM_MODULES'] - 1
            modules.append(HighResolutionModule(layer_config['NUM_BRANCHES'], block, layer_config['NUM_BLOCKS'],
                                                in_channels, layer_config['NUM_CHANNELS'],
                                                layer_config['FUSE_METHOD'], reset))
            in_channels = modules[-1].get_num_in_chs()
        return nn.Sequential(*modules), in_channels

    # ---- execution -------------------------------------------------------------------------------------
    @staticmethod
    def _transition(r, tr, src):
        if isinstance(tr[0], nn.Conv2d):          # a single Conv-BN-ReLU
            return _run_conv_bn(r, src, tr)
        for step in tr:                           # chain of stride-2 Conv-BN-ReLU
            src = _run_conv_bn(r, src, step)
        return src

    def _run(self, r, x: torch.Tensor):
        t = r.input(x, c_pad_to=4 if x.shape[1] <= 4 else 8)
        t = EF.conv_bn_act(r, t, self.conv1, self.bn1, relu=True)
        t = EF.conv_bn_act(r, t, self.conv2, self.bn2, relu=True)
        t = self.layer1(t)
        xl = [t if isinstance(tr, nn.Identity) else self._transition(r, tr, t) for tr in self.transition1]
        yl = self.stage2(xl)
        for trans, stage in ((self.transition2, self.stage3), (self.transition3, self.stage4)):
            xl = [yl[i] if isinstance(tr, nn.Identity) else self._transition(r, tr, yl[-1])
                  for i, tr in enumerate(trans)]
            yl = stage(xl)
        return yl

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        with engine.region() as r:
            yl = self._run(r, x)
            return list(r.output(*yl))

    def forward_features(self, x: torch.Tensor) -> List[torch.Tensor]:
        return [x] + self.forward(x)

    def get_stages(self, stage: int) -> nn.Module:
        output = [self.conv1, self.bn1, self.act1, self.conv2, self.bn2, self.act2]
        layers = [[self.layer1], [self.transition1, self.stage2], [self.transition2, self.stage3],
                  [self.transition3, self.stage4]]
        for i in range(stage):
            output += layers[i]
        return nn.ModuleList(output)


def _relu_inplace(seq: nn.Sequential) -> nn.Sequential:
    seq[2] = nn.ReLU(inplace=True)     # module identity only (parameter-free); execution is the fused unit
    return seq


def _create_hrnet(variant: str, pretrained: bool = False, **model_kwargs):
    for k in ('num_classes', 'global_pool', 'in_chans'):
        model_kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a download (no network here); pass '
                           f'pretrained=false and use task.load_checkpoint for local checkpoints')
    return HighResolutionNet(cfg_cls[variant], **model_kwargs)


def _register(variant):
    def entry(pretrained: bool = False, **kwargs):
        return _create_hrnet(variant, pretrained, **kwargs)
    entry.__name__ = variant
    entry.__doc__ = f"It's constructing a {variant} model."
    return BACKBONES.register_class(entry)


hrnet_w18_small = _register('hrnet_w18_small')
hrnet_w18_small_v2 = _register('hrnet_w18_small_v2')
hrnet_w18 = _register('hrnet_w18')
hrnet_w30 = _register('hrnet_w30')
hrnet_w32 = _register('hrnet_w32')
hrnet_w40 = _register('hrnet_w40')
hrnet_w44 = _register('hrnet_w44')
hrnet_w48 = _register('hrnet_w48')
hrnet_w64 = _register('hrnet_w64')
