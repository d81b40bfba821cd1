"""Teacher-forced per-unit parity against oracle/ (SURVEY.md §7 test plan; north_star: <= 1e-2 rel for bf16 activations
and gradients).

Every unit of the hot path is run ALONE on inputs taken from the oracle: the CPU oracle model runs one fp32 training step on a
seeded batch; forward hooks capture the input of each selected unit, tensor hooks the gradient of its output.  The unit is then
re-run three ways on the SAME (bf16-rounded) input / output-gradient and the SAME (bf16-exact) weights:

    oracle fp32          -- the parity target
    oracle bf16 autocast -- the reference's own `trainer.accelerator='cpu'` arithmetic under precision bf16,
                            printed as the yardstick of what bf16 storage costs
    HIP unit             -- torchok_amd's unit on libtok_gfx950.so

Two assertions per tensor (outputs, input gradients, every parameter gradient; relative L2): HIP vs the autocast oracle
<= 1e-2, and HIP vs the fp32 oracle <= max(1e-2, 1.25 x autocast-vs-fp32).  The second bound exceeds 1e-2 only where bf16
storage itself does: a ReLU behind a bf16-rounded pre-activation flips ~1e-3 of its elements, 2-3 % of a gradient's norm
for the reference's own bf16 run as well; all three distances are printed.

`-m gpu` runs the units on the MI355X; the same cases run on the host stand-in in the CPU suite (logic check of the harness
and of tests/fake_backend.py against oracle/)."""
import copy

import pytest
import torch
import torch.nn as nn

import oracle.hrnet_ref as H
import oracle.swin_ref as S
import oracle.timm_min as TM
import oracle.torchok_ref as R
import torchok_amd as T
from helpers import deterministic_state, record_distance, rel_err
from torchok_amd import engine
from torchok_amd.engine import functional as EF
from torchok_amd.engine import resample as ER
from torchok_amd.models.backbones import hrnet as PH
from torchok_amd.models.backbones import resnet as PR
from torchok_amd.models.backbones import swin as PS

TOL = 1e-2


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _bf(t):
    return t.detach().to(torch.bfloat16).float()


def _round_weights_(module):
    """Matrix / filter weights on the bf16 grid: both sides multiply exactly the same operand values."""
    with torch.no_grad():
        for p in module.parameters():
            if p.dim() >= 2:
                p.copy_(_bf(p))


def _capture(model, run, names):
    """Run `run()` (forward + backward of the oracle) and return {name: (input, grad_output)} of the named sub-modules."""
    mods = dict(model.named_modules())
    got, hooks = {}, []
    for n in names:
        def fwd(m, inp, out, n=n):
            o = out[0] if isinstance(out, (tuple, list)) else out
            got[n] = [inp[0].detach().clone() if torch.is_tensor(inp[0]) else [t.detach().clone() for t in inp[0]], None]
            if o.requires_grad:
                o.register_hook(lambda g, n=n: got[n].__setitem__(1, g.detach().clone()))
        hooks.append(mods[n].register_forward_hook(fwd))
    run()
    for h in hooks:
        h.remove()
    return got


def _ref_unit(unit, x, gout, autocast=False):
    """Oracle unit alone: (out, dx, {param: grad})."""
    unit = copy.deepcopy(unit).train()
    xs = [t.clone().requires_grad_(True) for t in (x if isinstance(x, (list, tuple)) else [x])]
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
        out = unit(xs if isinstance(x, (list, tuple)) else xs[0])
    outs = list(out) if isinstance(out, (list, tuple)) else [out]
    gouts = gout if isinstance(gout, (list, tuple)) else [gout]
    torch.autograd.backward([o.float() for o in outs], [g for g in gouts])
    return [o.detach().float() for o in outs], [t.grad for t in xs], {n: p.grad for n, p in unit.named_parameters()}


def _report(tag, what, e_hip, e_yard, e_pair):
    record_distance(f'units/{tag}', what, hip_vs_autocast=e_pair, hip_vs_fp32=e_hip, autocast_vs_fp32=e_yard)
    print(f'[unit {tag}] {what:34s} HIP-vs-autocast {e_pair:.2e}   HIP-vs-fp32 {e_hip:.2e}   autocast-vs-fp32 {e_yard:.2e}')


def _check(tag, ours, ref32, ref_ac, composite=False):
    """ours / ref32 / ref_ac: (outs, dxs, param grads).  Two gates per tensor (relative L2):
      (1) HIP vs the bf16-autocast oracle  <= 1e-2  — the reference's own arithmetic under `precision: bf16`, same rounding
          points (conv / linear outputs and normalised activations are bf16 tensors), teacher-forced inputs — or HIP vs the
          fp32 oracle <= 1e-2 outright;
      (2) HIP vs the fp32 oracle <= max(1e-2, 1.25 x autocast-vs-fp32) — no further from exact arithmetic than bf16 storage
          itself costs (a ReLU behind a bf16-rounded pre-activation flips ~1e-3 of the elements: 2-3 % of a gradient's norm
          for ANY bf16 pipeline, the reference's included; the yardstick is printed).
    composite=True (a whole residual block / HR module): gate (1) is applied to the outputs only.  Inside such a unit torch
    rounds `bn3(y)` to bf16 BEFORE adding the shortcut while the fused kernel adds in fp32 and rounds once: two valid bf16
    evaluations of one fp32 function whose ReLU decisions differ on ~1e-4 of the elements (1-2 % of a gradient norm);
    gradients of composite units are therefore held to gate (2) — with 1.5 x the yardstick: which elements flip is a coin
    toss per implementation, the two flip sets are independent samples of the same size —, single units to both."""
    def gate(what, o, r32, rac, grad=True):
        e, y, pair = rel_err(o, r32), rel_err(rac, r32), rel_err(o, rac)
        _report(tag, what, e, y, pair)
        if not (composite and grad):
            # (within 1e-2 of the fp32 oracle is better still: where the autocast run is itself ~1e-2 from fp32 — bf16
            #  LayerNorm statistics — its distance to an fp32-accurate result says nothing about that result)
            assert pair < TOL or e < TOL, (tag, what, 'vs autocast oracle', pair, 'vs fp32 oracle', e)
        assert e < max(TOL, (1.5 if composite else 1.25) * y), (tag, what, 'vs fp32 oracle', e, y)
    for i, (o, r32, rac) in enumerate(zip(ours[0], ref32[0], ref_ac[0])):
        gate(f'out[{i}]', o, r32, rac.float(), grad=False)
    for i, (o, r32, rac) in enumerate(zip(ours[1], ref32[1], ref_ac[1])):
        if r32 is not None and o is not None:      # (the 3-channel image takes no gradient)
            gate(f'd(input[{i}])', o, r32, rac.float())
    for n, r32 in ref32[2].items():
        if r32 is None or float(r32.norm()) < 1e-6 * max(1.0, float(r32.numel()) ** 0.5):
            continue        # a gradient that is analytically zero (e.g. a conv bias in front of BatchNorm)
        gate(f'd({n})', ours[2][n], r32, ref_ac[2][n].float())


def _to_dev_nhwc(x, dev, grad=True):
    t = x.to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return t.requires_grad_(grad)


def _ours_map(build, x, gout, dev, params_of):
    """Run a product unit on 4-D maps: build(region, [TTensor]) -> TTensor | list."""
    xs = [_to_dev_nhwc(t, dev, grad=t.shape[1] % 8 == 0) for t in (x if isinstance(x, (list, tuple)) else [x])]
    with engine.region() as r:
        ins = [r.input(t, c_pad_to=4 if t.shape[1] <= 4 else 8) for t in xs]
        out = build(r, ins)
        outs = out if isinstance(out, (list, tuple)) else [out]
        res = r.output(*outs)
    res = list(res) if isinstance(res, (tuple, list)) else [res]
    gouts = gout if isinstance(gout, (list, tuple)) else [gout]
    torch.autograd.backward(res, [g.to(dev).to(res[i].dtype) for i, g in enumerate(gouts)])
    if dev == 'cuda':
        torch.cuda.synchronize()
    return ([o.detach().float().cpu() for o in res], [None if t.grad is None else t.grad.float().cpu() for t in xs],
            {n: (None if p.grad is None else p.grad.detach().float().cpu()) for n, p in params_of.named_parameters()})


# =================================================================================================================
# ResNet-50 units: every distinct conv-bn-relu geometry and whole bottlenecks, inputs from the oracle's activations
# =================================================================================================================

@pytest.fixture(scope='module')
def resnet50_capture():
    torch.manual_seed(0)
    ref = R.ClassificationModel('resnet50', 16, zero_init_last=False).train()
    ref.load_state_dict(deterministic_state(ref.state_dict(), 31))
    _round_weights_(ref)
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(8, 3, 128, 128, generator=g))
    y = torch.randint(0, 16, (8,), generator=g)
    names = ['backbone.layer1.0', 'backbone.layer1.1', 'backbone.layer2.0', 'backbone.layer2.1', 'backbone.layer3.0',
             'backbone.layer3.2', 'backbone.layer4.0', 'backbone.layer4.2',
             'backbone.layer1.1.conv1', 'backbone.layer1.1.conv2', 'backbone.layer1.1.conv3',
             'backbone.layer2.0.conv2', 'backbone.layer2.0.downsample', 'backbone.layer3.1.conv2', 'backbone.layer4.0.conv2']

    def run():
        out = ref.forward_with_gt({'image': x, 'target': y})
        nn.functional.cross_entropy(out['prediction'], y).backward()
    cap = _capture(ref, run, names)
    return ref, cap


class _CBR(nn.Module):
    """conv -> bn -> relu of the oracle, as one unit (what torchok_amd's conv_bn_act replaces)."""

    def __init__(self, conv, bn, relu=True):
        super().__init__()
        self.conv, self.bn, self.relu = conv, bn, relu

    def forward(self, x):
        y = self.bn(self.conv(x))
        return torch.relu(y) if self.relu else y


@pytest.mark.parametrize('conv,bn,relu', [('layer1.1.conv1', 'layer1.1.bn1', True),      # 1x1 256 -> 64
                                          ('layer1.1.conv2', 'layer1.1.bn2', True),      # 3x3 64 -> 64
                                          ('layer2.0.conv2', 'layer2.0.bn2', True),      # 3x3 stride 2
                                          ('layer2.0.downsample.0', 'layer2.0.downsample.1', False),   # 1x1 stride 2, no act
                                          ('layer3.1.conv2', 'layer3.1.bn2', True),      # 3x3 256 -> 256 (deep K)
                                          ('layer4.0.conv2', 'layer4.0.bn2', True)])     # 3x3 stride 2 at 8 -> 4 px
def test_conv_bn_relu_unit(dev, resnet50_capture, conv, bn, relu):
    ref, cap = resnet50_capture
    mods = dict(ref.backbone.named_modules())
    unit = _CBR(mods[conv], mods[bn], relu)
    key = 'backbone.' + (conv if not conv.endswith('.0') else conv[:-2])
    x = _bf(cap[key][0])
    # the captured gradient belongs to the conv output; the unit's output gradient is drawn at the same scale
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        shape = unit(x).shape
    scale = float(cap['backbone.layer1.1'][1].abs().mean())
    gout = _bf(torch.randn(shape, generator=g) * scale)
    r32, rac = _ref_unit(unit, x, gout), _ref_unit(unit, x, gout, autocast=True)
    ours_unit = copy.deepcopy(unit).to(dev).train()
    ours = _ours_map(lambda r, ins: EF.conv_bn_act(r, ins[0], ours_unit.conv, ours_unit.bn, relu=relu), x, gout, dev, ours_unit)
    _check(f'{conv}', ours, r32, rac)
    # BatchNorm side effects: running statistics (momentum 0.1, unbiased variance) and the exact step counter
    chk = copy.deepcopy(unit).train()
    chk(x)
    assert rel_err(ours_unit.bn.running_mean, chk.bn.running_mean) < 1e-3
    assert rel_err(ours_unit.bn.running_var, chk.bn.running_var) < 2e-3
    assert int(ours_unit.bn.num_batches_tracked) == int(chk.bn.num_batches_tracked)


@pytest.mark.parametrize('fused', [True, False], ids=['fused-unit3', 'plain'])
@pytest.mark.parametrize('name', ['layer1.0', 'layer1.1', 'layer2.0', 'layer2.1', 'layer3.0', 'layer3.2', 'layer4.0',
                                  'layer4.2'])
def test_bottleneck_unit(dev, resnet50_capture, name, fused, monkeypatch):
    """[timm] Bottleneck (with / without the projection shortcut, stride 1 / 2) on the oracle's own block input and output
    gradient — through both execution plans of its residual unit: 'fused-unit3' (engine.functional._Unit3Node: conv3 / bn3 /
    add / ReLU and the stride-1 projection shortcut without their pre-BatchNorm tensors, the plan of the large maps) and
    'plain' (conv, BatchNorm statistics, apply as three launches)."""
    monkeypatch.setattr(EF, 'UNIT3_MIN_ROWS', 0 if fused else 1 << 40)
    ref, cap = resnet50_capture
    blk = dict(ref.backbone.named_modules())[name]
    x, gout = _bf(cap['backbone.' + name][0]), _bf(cap['backbone.' + name][1])
    r32, rac = _ref_unit(blk, x, gout), _ref_unit(blk, x, gout, autocast=True)
    ds = None
    if blk.downsample is not None:
        ds = nn.Sequential(copy.deepcopy(blk.downsample[0]), copy.deepcopy(blk.downsample[1]))
    ours_blk = PR.Bottleneck(blk.conv1.in_channels, blk.conv3.out_channels // 4, stride=blk.conv2.stride[0], downsample=ds)
    ours_blk.load_state_dict(blk.state_dict())
    ours_blk.to(dev).train()
    ours = _ours_map(lambda r, ins: ours_blk(ins[0]), x, gout, dev, ours_blk)
    _check(f'bottleneck {name}', ours, r32, rac, composite=True)


def test_stem_unit(dev, resnet50_capture):
    """7x7/s2 conv (3 -> 64, the c4 operand path) + BN + ReLU + 3x3/s2 max-pool as ONE fused unit; input = the image."""
    ref, _ = resnet50_capture
    bb = ref.backbone

    class Stem(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1, self.maxpool = bb.conv1, bb.bn1, bb.maxpool

        def forward(self, x):
            return self.maxpool(torch.relu(self.bn1(self.conv1(x))))
    unit = Stem()
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(8, 3, 128, 128, generator=g))
    with torch.no_grad():
        shape = unit(x).shape
    gout = _bf(torch.randn(shape, generator=g) * 1e-3)
    r32, rac = _ref_unit(unit, x, gout), _ref_unit(unit, x, gout, autocast=True)
    ours_unit = copy.deepcopy(unit).to(dev).train()
    ours = _ours_map(lambda r, ins: EF.conv_bn_act(r, ins[0], ours_unit.conv1, ours_unit.bn1, relu=True, pool=True),
                     x, gout, dev, ours_unit)
    _check('stem', ours, r32, rac)


# =================================================================================================================
# SwinV2 units at the real geometry (window 7, heads of 32 channels), inputs from the oracle's activations
# =================================================================================================================

@pytest.fixture(scope='module')
def swin_capture():
    torch.manual_seed(0)
    ref = S.SwinV2(img_size=112, window_size=7, embed_dim=96, depths=(2, 2, 2), num_heads=(3, 6, 12),
                   drop_path_rate=0.0).train()
    ref.load_state_dict(deterministic_state(ref.state_dict(), 33), strict=False)
    _round_weights_(ref)
    g = torch.Generator().manual_seed(6)
    x = _bf(torch.randn(4, 3, 112, 112, generator=g))
    names = ['layers.0.blocks.0', 'layers.0.blocks.1', 'layers.1.blocks.1', 'layers.2.blocks.0', 'layers.0.downsample',
             'layers.1.downsample', 'patch_embed']

    def run():
        ref(x).square().mean().backward()
    return ref, _capture(ref, run, names), x


def _ours_tokens(run, x, gout, dev, params_of):
    """Product token units: x (B, L, C) -> rows (B*L, C)."""
    b, l, c = x.shape
    xt = x.reshape(b * l, c).to(dev).to(torch.bfloat16).requires_grad_(True)
    with engine.region() as r:
        out = r.output(run(r, r.input(xt), b))
    out.backward(gout.reshape(-1, gout.shape[-1]).to(dev).to(out.dtype))
    if dev == 'cuda':
        torch.cuda.synchronize()
    return ([out.detach().float().cpu().reshape(gout.shape)], [xt.grad.float().cpu().reshape(x.shape)],
            {n: (None if p.grad is None else p.grad.detach().float().cpu()) for n, p in params_of.named_parameters()})


@pytest.mark.parametrize('name,res,dim,heads,shift', [('layers.0.blocks.0', 28, 96, 3, 0), ('layers.0.blocks.1', 28, 96, 3, 3),
                                                      ('layers.1.blocks.1', 14, 192, 6, 3), ('layers.2.blocks.0', 7, 384, 12, 0)])
def test_swin_block_unit(dev, swin_capture, name, res, dim, heads, shift):
    """SwinTransformerBlock (res-post-norm): cosine window attention w7 (+ shift mask, continuous position bias), LN-residual,
    MLP, LN-residual — on the oracle's own block input and output gradient."""
    ref, cap, _ = swin_capture
    blk = dict(ref.named_modules())[name]
    x, gout = _bf(cap[name][0]), _bf(cap[name][1])
    r32, rac = _ref_unit(blk, x, gout), _ref_unit(blk, x, gout, autocast=True)
    ours_blk = PS.SwinTransformerBlock(dim=dim, input_resolution=(res, res), num_heads=heads, window_size=7,
                                       shift_size=shift, drop_path=0.0)
    missing = ours_blk.load_state_dict(blk.state_dict(), strict=False)
    assert not missing.missing_keys, missing
    assert tuple(ours_blk.shift_size) == ((shift, shift) if res > 7 else (0, 0))
    ours_blk.to(dev).train()
    ours = _ours_tokens(lambda r, t, b: ours_blk.run(r, t, b), x, gout, dev, ours_blk)
    # composite: the continuous-position-bias MLP holds a ReLU behind a bf16 Linear (the flip argument of _check)
    _check(f'swin block {name} (shift {shift})', ours, r32, rac, composite=True)


@pytest.mark.parametrize('name,res,dim', [('layers.0.downsample', 28, 96), ('layers.1.downsample', 14, 192)])
def test_patch_merging_unit(dev, swin_capture, name, res, dim):
    ref, cap, _ = swin_capture
    pm = dict(ref.named_modules())[name]
    x, gout = _bf(cap[name][0]), _bf(cap[name][1])
    r32, rac = _ref_unit(pm, x, gout), _ref_unit(pm, x, gout, autocast=True)
    ours_pm = PS.PatchMerging((res, res), dim)
    ours_pm.load_state_dict(pm.state_dict())
    ours_pm.to(dev).train()
    ours = _ours_tokens(lambda r, t, b: ours_pm.run(r, t, b), x, gout, dev, ours_pm)
    _check(f'patch merging {name}', ours, r32, rac)


def test_patch_embed_unit(dev, swin_capture):
    ref, cap, x = swin_capture
    pe = ref.patch_embed
    gout = _bf(cap['patch_embed'][1])
    r32, rac = _ref_unit(pe, x, gout), _ref_unit(pe, x, gout, autocast=True)
    ours_pe = PS.PatchEmbed(img_size=112, patch_size=4, in_chans=3, embed_dim=96, norm_layer=nn.LayerNorm)
    ours_pe.load_state_dict(pe.state_dict())
    ours_pe.to(dev).train()
    img = x.to(dev)
    with engine.region() as r:
        out = r.output(ours_pe.run(r, img))
    out.backward(gout.reshape(-1, 96).to(dev).to(out.dtype))
    ours = ([out.detach().float().cpu().reshape(gout.shape)], [None],
            {n: p.grad.detach().float().cpu() for n, p in ours_pe.named_parameters()})
    _check('patch embed', ours, (r32[0], [None], r32[2]), (rac[0], [None], rac[2]))


# =================================================================================================================
# HRNet units: all-to-all fuse (nearest-up / strided chains, sum, ReLU), bilinear concat neck, segmentation head + CE
# =================================================================================================================

@pytest.fixture(scope='module')
def hrnet_capture():
    torch.manual_seed(0)
    ref = H.SegmentationModel('hrnet_w18_small', 19).train()
    ref.load_state_dict(deterministic_state(ref.state_dict(), 35))
    _round_weights_(ref)
    g = torch.Generator().manual_seed(7)
    x = _bf(torch.randn(2, 3, 128, 256, generator=g))
    y = torch.randint(0, 19, (2, 128, 256), generator=g)
    y[:, :3] = 255
    got = {}

    def run():
        feats = ref.backbone.forward_features(x)
        for i, f in enumerate(feats[1:]):
            f.retain_grad()
        got['feats'] = feats
        neck = ref.neck_forward(feats)
        neck[1].retain_grad()
        got['neck'] = neck[1]
        logits = ref.head_forward(neck)
        logits.retain_grad()
        got['logits'] = logits
        nn.functional.cross_entropy(logits, y, ignore_index=255).backward()
    cap = _capture(ref, run, ['backbone.stage3.0', 'backbone.stage4.0'])
    return ref, cap, got, x, y


@pytest.mark.parametrize('name,nb', [('backbone.stage3.0', 3), ('backbone.stage4.0', 4)])
def test_hr_fuse_unit(dev, hrnet_capture, name, nb):
    """The all-to-all fuse of [timm] HighResolutionModule — per output branch: 1x1 conv-BN + nearest upsample of the lower
    resolutions, strided 3x3 conv-BN(-ReLU) chains of the higher ones, sum, ReLU — on the oracle's branch maps (the
    BasicBlock branches are covered by the block units above and are replaced by the identity on both sides); list of maps
    in, list of maps out, every output gets a gradient."""
    ref, cap, _, _, _ = hrnet_capture
    mod = copy.deepcopy(dict(ref.named_modules())[name])
    mod.branches = nn.ModuleList(nn.Identity() for _ in range(nb))
    xs = [_bf(t) for t in cap[name][0]]
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        shapes = [o.shape for o in copy.deepcopy(mod)([t.clone() for t in xs])]
    gouts = [_bf(torch.randn(s, generator=g) * 1e-3) for s in shapes]
    r32, rac = _ref_unit(mod, xs, gouts), _ref_unit(mod, xs, gouts, autocast=True)
    chs = [t.shape[1] for t in xs]
    ours = PH.HighResolutionModule(nb, PR.BasicBlock, [2] * nb, list(chs), list(chs), 'SUM')
    ours.branches = nn.ModuleList(nn.Identity() for _ in range(nb))
    ours.load_state_dict(mod.state_dict())
    ours.to(dev).train()
    res = _ours_map(lambda r, ins: ours(list(ins)), xs, gouts, dev, ours)
    _check(f'hr fuse {name}', res, r32, rac, composite=True)


def test_hr_neck_and_head_units(dev, hrnet_capture):
    """HRNetSegmentationNeck (bilinear x3 -> concat -> 1x1 ConvBnReLU) and SegmentationHead (1x1 conv + bias -> bilinear to
    the image size) followed by the pixel-wise CrossEntropyLoss (ignore_index 255), on the oracle's feature maps."""
    ref, _, got, x, y = hrnet_capture
    feats = [x] + [_bf(f) for f in got['feats'][1:]]

    class Neck(nn.Module):
        def __init__(self):
            super().__init__()
            self.neck = ref.neck

        def forward(self, fs):
            return ref.__class__.neck_forward(self, [x] + list(fs))[1]
    neck = Neck()
    gneck = _bf(got['neck'].grad)
    r32, rac = _ref_unit(neck, feats[1:], gneck), _ref_unit(neck, feats[1:], gneck, autocast=True)
    ours_neck = T.NECKS.get('HRNetSegmentationNeck')(in_channels=[f.shape[1] for f in feats[1:]])
    ours_neck.load_state_dict(ref.neck.state_dict())
    ours_neck.to(dev).train()
    xs = [_to_dev_nhwc(f, dev) for f in feats[1:]]
    out = ours_neck([x.to(dev)] + xs)[1]
    out.backward(gneck.to(dev).to(out.dtype))
    ours = ([out.detach().float().cpu()], [t.grad.float().cpu() for t in xs],
            {'neck.' + n: p.grad.detach().float().cpu() for n, p in ours_neck.named_parameters()})
    # composite: the neck runs in the commuted order (engine/neck.py: the pointwise product at every source's own resolution,
    # THEN interpolate and sum) — the same fp32 function as interpolate -> concat -> product with other bf16 rounding points
    # (four partial products instead of four interpolated inputs), so its gradients are held to the fp32 yardstick, its
    # output to both gates
    _check('hr seg neck', ours, r32, rac, composite=True)

    # head + loss: logits and the loss value and d(features)
    class HeadLoss(nn.Module):
        def __init__(self):
            super().__init__()
            self.head, self.num_classes = ref.head, 19

        def forward(self, f):
            logits = ref.__class__.head_forward(self, [x, f])
            return nn.functional.cross_entropy(logits.float(), y, ignore_index=255), logits
    hl = HeadLoss()
    f_in = _bf(got['neck'])
    outs = {}
    for tag, ac in (('fp32', False), ('ac', True)):
        m = copy.deepcopy(hl)
        fi = f_in.clone().requires_grad_(True)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=ac):
            loss, logits = m(fi)
        loss.backward()
        outs[tag] = (float(loss), logits.detach().float(), fi.grad, {n: p.grad for n, p in m.named_parameters()})
    ours_head = T.HEADS.get('SegmentationHead')(in_channels=f_in.shape[1], num_classes=19)
    ours_head.load_state_dict(ref.head.state_dict())
    ours_head.to(dev).train()
    fi = _to_dev_nhwc(f_in, dev)
    logits = ours_head([x.to(dev), fi])
    loss = T.LOSSES.get('CrossEntropyLoss')(ignore_index=255)(input=logits, target=y.to(dev))
    loss.backward()
    e_loss = abs(float(loss.detach()) - outs['fp32'][0]) / abs(outs['fp32'][0])
    print(f'[unit hr head+CE] loss HIP {float(loss.detach()):.6f} fp32 {outs["fp32"][0]:.6f} autocast {outs["ac"][0]:.6f}')
    assert e_loss < 2e-3
    res = ([logits.detach().float().cpu()], [fi.grad.float().cpu()],
           {'head.' + n: p.grad.detach().float().cpu() for n, p in ours_head.named_parameters()})
    _check('hr seg head + CE', res, ([outs['fp32'][1]], [outs['fp32'][2]], outs['fp32'][3]),
           ([outs['ac'][1]], [outs['ac'][2]], outs['ac'][3]))


# =================================================================================================================
# classification tail: global average pool -> Linear -> CrossEntropyLoss at the 1000-class size
# =================================================================================================================

def test_pool_head_ce_unit(dev):
    g = torch.Generator().manual_seed(13)
    x = _bf(torch.randn(32, 2048, 7, 7, generator=g).relu())
    y = torch.randint(0, 1000, (32,), generator=g)
    fc = nn.Linear(2048, 1000)
    with torch.no_grad():
        fc.weight.copy_(_bf(torch.randn(1000, 2048, generator=g) * 0.02))
        fc.bias.copy_(torch.randn(1000, generator=g) * 0.1)
    outs = {}
    for tag, ac in (('fp32', False), ('ac', True)):
        m = copy.deepcopy(fc)
        xi = x.clone().requires_grad_(True)
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=ac):
            logits = m(torch.flatten(nn.functional.adaptive_avg_pool2d(xi, 1), 1))
        loss = nn.functional.cross_entropy(logits.float(), y)
        loss.backward()
        outs[tag] = (float(loss), logits.detach().float(), xi.grad, {'fc.' + n: p.grad for n, p in m.named_parameters()})
    pool = T.POOLINGS.get('Pooling')(in_channels=2048).to(dev)
    head = T.HEADS.get('ClassificationHead')(in_channels=2048, num_classes=1000)
    head.fc.load_state_dict(fc.state_dict())
    head.to(dev).train()
    xi = _to_dev_nhwc(x, dev)
    logits = head(pool(xi))
    loss = T.LOSSES.get('CrossEntropyLoss')()(input=logits, target=y.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - outs['fp32'][0]) < 2e-3 * outs['fp32'][0]
    res = ([logits.detach().float().cpu()], [xi.grad.float().cpu()],
           {n: p.grad.detach().float().cpu() for n, p in head.named_parameters()})
    _check('pool + fc + CE', res, ([outs['fp32'][1]], [outs['fp32'][2]], outs['fp32'][3]),
           ([outs['ac'][1]], [outs['ac'][2]], outs['ac'][3]))
