"""HRNet segmentation rows (SURVEY.md §8 a12-a14 + a10 on (N,C,H,W) logits): HighResolutionNet,
HRNetSegmentationNeck, SegmentationHead, SegmentationTask against tests/golden/hrnet_seg_step.npz (one training
step of the reference's own hrnet.py / neck / head, tests/golden/gen_golden.py) and oracle/hrnet_ref.py.
Each test runs on the host stand-in and, marked gpu, through libtok_gfx950.so."""
import copy
import os

import numpy as np
import pytest
import torch

import oracle.hrnet_ref as H
import torchok_amd as T
from helpers import deterministic_state, rel_err
from torchok_amd.constructor.config import apply_schema

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'hrnet_seg_step.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def seg_config(backbone='hrnet_w18_small', classes=19, size=128):
    return apply_schema({
        'task': {'name': 'SegmentationTask',
                 'params': {'backbone_name': backbone, 'backbone_params': {'pretrained': False, 'in_channels': 3},
                            'neck_name': 'HRNetSegmentationNeck', 'head_name': 'SegmentationHead',
                            'head_params': {'num_classes': classes},
                            'inputs': [{'shape': [3, size, size], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': 'CrossEntropyLoss', 'params': {'ignore_index': 255},
                                   'mapping': {'input': 'prediction', 'target': 'target'}}]},
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.01, 'momentum': 0.9, 'weight_decay': 5e-4}}}],
        'data': {}, 'trainer': {'precision': 'bf16'}})


def _inputs():
    return torch.from_numpy(GOLD['x'].astype(np.float32)), torch.from_numpy(GOLD['y'].astype(np.int64))


def test_oracle_reproduces_reference_step():
    x, y = _inputs()
    ora = H.SegmentationModel(str(GOLD['variant']), int(GOLD['num_classes'])).train()
    ora.load_state_dict(deterministic_state(ora.state_dict(), int(GOLD['seed'])))
    assert [n for n, _ in ora.named_parameters()] == [str(n) for n in GOLD['param_names']]
    out = ora.forward_with_gt({'image': x, 'target': y})
    loss = torch.nn.functional.cross_entropy(out['prediction'], y, ignore_index=255)
    assert float(loss) == float(GOLD['loss'])
    assert np.array_equal(out['prediction'].detach().half().numpy(), GOLD['prediction'])
    loss.backward()
    gn = np.array([float(p.grad.double().norm()) for p in ora.parameters()])
    assert np.allclose(gn, GOLD['grad_norm'], rtol=1e-6)
    assert sum(p.numel() for p in H.SegmentationModel('hrnet_w48', 19).parameters()) == 65858659


def test_reference_shape_tests(dev):
    """tests/additional_tests/models/backbones/test_backbone.py:75-89 (hrnet_w18_small on 2x3x64x64)."""
    m = T.BACKBONES.get('hrnet_w18_small')(pretrained=False).to(dev)
    x = torch.rand(2, 3, 64, 64).to(dev)
    with torch.no_grad():
        out = m(x)
    assert [tuple(f.shape) for f in out] == [(2, 16, 16, 16), (2, 32, 8, 8), (2, 64, 4, 4), (2, 128, 2, 2)]
    feats = m.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(2, 3, 64, 64), (2, 16, 16, 16), (2, 32, 8, 8), (2, 64, 4, 4),
                                               (2, 128, 2, 2)]
    assert m.out_encoder_channels == (16, 32, 64, 128)
    assert T.NECKS.get('HRNetSegmentationNeck')((18, 36, 72, 144)).out_channels == 270    # necks/test_hrnet.py:26
    assert len(m.get_stages(4)) == 6 + 1 + 2 + 2 + 2
    # state_dict layout = reference layout (timm names)
    ref = H.HRNet('hrnet_w18_small')
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    with pytest.raises(ValueError, match='divisible by 32'):
        m(torch.rand(1, 3, 72, 72).to(dev))


def test_segmentation_step_vs_reference_golden(dev):
    cfg = seg_config()
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')},
                             int(GOLD['seed']))
    task.load_state_dict(sd, strict=False)
    task.to(dev).train()
    x, y = _inputs()
    feats = task.backbone.forward_features(x.to(dev))
    assert [list(f.shape) for f in feats[1:]] == GOLD['feat_shapes'].tolist()
    for f, ss in zip(feats[1:], GOLD['feat_sumsq']):
        assert abs(float((f.detach().double() ** 2).sum()) / float(ss) - 1) < 3e-2
    out = task.training_step({'image': x.to(dev), 'target': y.to(dev)}, 0)
    fw = task.forward_with_gt({'image': x.to(dev), 'target': y.to(dev)})
    assert set(fw) == {'prediction', 'target'} and fw['prediction'].shape == (4, 19, 128, 128)
    pred = fw['prediction'].detach().float().cpu()
    assert rel_err(pred, torch.from_numpy(GOLD['prediction'].astype(np.float32))) < 3e-2
    assert abs(float(out['loss'].detach()) - float(GOLD['loss'])) < 2e-2 * float(GOLD['loss'])
    out['loss'].backward()
    names = [str(n) for n in GOLD['param_names']]
    assert names == [n for n, _ in task.named_parameters()]

    # bf16 yardstick: torch's bf16-autocast CPU run of the oracle on the same weights
    ora = H.SegmentationModel(str(GOLD['variant']), 19).train()
    ora.load_state_dict(sd)
    ref32 = copy.deepcopy(ora)
    torch.nn.functional.cross_entropy(ref32.forward_with_gt({'image': x, 'target': y})['prediction'], y,
                                      ignore_index=255).backward()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o = ora.forward_with_gt({'image': x, 'target': y})
    torch.nn.functional.cross_entropy(o['prediction'].float(), y, ignore_index=255).backward()
    g32 = {n: p.grad for n, p in ref32.named_parameters()}
    yard = np.array([rel_err(p.grad, g32[n]) for n, p in ora.named_parameters()])
    errs = np.array([rel_err(p.grad, g32[n]) for n, p in task.named_parameters()])
    assert np.median(errs) < 1.5 * np.median(yard) + 1e-2, (np.median(errs), np.median(yard))
    assert (errs < 1.5 * yard + 0.1).mean() > 0.9, np.sort(errs - 1.5 * yard)[-5:]
    gn = np.array([float(p.grad.detach().double().norm()) for _, p in task.named_parameters()])
    assert np.median(np.abs(gn / GOLD['grad_norm'] - 1)) < 0.1
    opt = task.configure_optimizers()[0]['optimizer']
    before = {n: p.detach().clone() for n, p in task.named_parameters()}
    opt.step()
    assert all(not torch.equal(before[n], p.detach()) for n, p in task.named_parameters())


def test_eval_forward_and_single_class(dev):
    cfg = seg_config(classes=1, size=64)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).to(dev).eval()
    with torch.no_grad():
        y = task(torch.rand(2, 3, 64, 64).to(dev))
    assert y.shape == (2, 64, 64)                    # heads/segmentation/base.py:38-39 squeeze
    assert len(task.as_module()) == 3


def test_padded_channel_variant_step(dev):
    """hrnet_w18_small_v2: branch widths 18/36/72/144 (not multiples of 8) -> zero-padded activations, BatchNorm over
    num_features < padded width, concat at logical offsets (270 -> pitch 272).  Step vs the fp32 oracle."""
    cfg = seg_config('hrnet_w18_small_v2', classes=5, size=64)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 7)
    task.load_state_dict(sd, strict=False)
    task.to(dev).train()
    ora = H.SegmentationModel('hrnet_w18_small_v2', 5).train()
    ora.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 3, 64, 64, generator=g)
    y = torch.randint(0, 5, (4, 64, 64), generator=g)
    feats = task.backbone.forward_features(x.to(dev))
    assert [tuple(f.shape) for f in feats[1:]] == [(4, 18, 16, 16), (4, 36, 8, 8), (4, 72, 4, 4), (4, 144, 2, 2)]
    neck_out = task.neck(feats)
    assert neck_out[1].shape == (4, 270, 16, 16)
    out = task.training_step({'image': x.to(dev), 'target': y.to(dev)}, 0)
    o32 = ora.forward_with_gt({'image': x, 'target': y})
    l32 = torch.nn.functional.cross_entropy(o32['prediction'], y, ignore_index=255)
    assert abs(float(out['loss'].detach()) - float(l32.detach())) < 3e-2 * float(l32.detach())
    out['loss'].backward()
    l32.backward()
    ac = copy.deepcopy(ora)
    ac.zero_grad()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        oa = ac.forward_with_gt({'image': x, 'target': y})
    torch.nn.functional.cross_entropy(oa['prediction'].float(), y, ignore_index=255).backward()
    g32 = {n: p.grad for n, p in ora.named_parameters()}
    yard = np.array([rel_err(p.grad, g32[n]) for n, p in ac.named_parameters()])
    errs = np.array([rel_err(p.grad, g32[n]) for n, p in task.named_parameters()])
    assert all(p.grad.shape == p.shape for p in task.parameters())
    assert np.median(errs) < 1.5 * np.median(yard) + 1e-2, (np.median(errs), np.median(yard))


def test_reference_neck_test_w18(dev):
    """tests/additional_tests/models/necks/test_hrnet.py:20-26 of the reference (hrnet_w18, 2x3x224x224)."""
    backbone = T.BACKBONES.get('hrnet_w18')(pretrained=False).to(dev)
    neck = T.NECKS.get('HRNetSegmentationNeck')(backbone.out_encoder_channels).to(dev)
    x = torch.rand(2, 3, 224, 224).to(dev)
    with torch.no_grad():
        input_image, features = neck(backbone.forward_features(x))
    assert tuple(features.shape) == (2, 270, 56, 56)
    assert tuple(input_image.shape) == (2, 3, 224, 224)


def test_segmentation_neck_commuted_vs_direct(dev, monkeypatch):
    """engine/neck.py: conv1x1(cat_j up(x_j)) evaluated as sum_j up(conv1x1_j(x_j)) — against the direct order (interpolate
    into the concat buffer, one product) on the same inputs and parameters: output, input gradients, parameter gradients and
    the BatchNorm running statistics agree to bf16 rounding; state_dict names are those of the reference either way."""
    from torchok_amd.engine import neck as EN
    chans = (16, 32, 64, 128)
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(3, c, 32 >> i, 48 >> i, generator=g) for i, c in enumerate(chans)]
    img = torch.zeros(3, 3, 128, 192)
    gout = torch.randn(3, sum(chans), 32, 48, generator=g)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(EN, 'NECK_COMMUTE', mode)
        neck = T.NECKS.get('HRNetSegmentationNeck')(chans)
        neck.load_state_dict(deterministic_state(neck.state_dict(), 11))
        neck.to(dev).train()
        assert set(neck.state_dict()) == {'convbnact.conv.weight', 'convbnact.bn.weight', 'convbnact.bn.bias',
                                          'convbnact.bn.running_mean', 'convbnact.bn.running_var',
                                          'convbnact.bn.num_batches_tracked'}
        xd = [t.to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True) for t in xs]
        out = neck([img.to(dev)] + xd)[1]
        out.backward(gout.to(dev).to(out.dtype))
        res[mode] = ([out.detach().float().cpu()] + [t.grad.float().cpu() for t in xd]
                     + [p.grad.float().cpu() for p in neck.parameters()]
                     + [neck.convbnact.bn.running_mean.float().cpu(), neck.convbnact.bn.running_var.float().cpu()])
        assert int(neck.convbnact.bn.num_batches_tracked) == 1
    names = ['out'] + [f'd(x{i})' for i in range(4)] + ['d(conv.weight)', 'd(bn.weight)', 'd(bn.bias)', 'running_mean',
                                                      'running_var']
    for nm, a, b in zip(names, res[True], res[False]):
        assert a.shape == b.shape
        # gradients behind a ReLU on a bf16-rounded pre-activation: the two orders round y differently (5e-3 apart), ~0.3 % of the
        # ReLU decisions flip, 2-4 % of a gradient's norm between ANY two bf16 evaluations (tests/test_units_gpu.py, gate 2)
        assert rel_err(a, b) < (6e-2 if nm.startswith('d(') else 2e-2), (nm, rel_err(a, b))


def test_classification_neck(dev):
    """HRNetClassificationNeck: reference shape test (necks/test_hrnet.py:15-19, hrnet_w18 -> (2, 2048, 7, 7)) and values /
    gradients / BatchNorm side effects vs the oracle restatement (the loop overwrites y, as in the reference)."""
    chans = (16, 32, 64, 128)
    neck = T.NECKS.get('HRNetClassificationNeck')(chans)
    ref = H.ClassificationNeck(chans)
    assert {k: tuple(v.shape) for k, v in neck.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    sd = deterministic_state(ref.state_dict(), 9)
    ref.load_state_dict(sd)
    neck.load_state_dict(sd)
    neck.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, c, 32 >> i, 32 >> i, generator=g) for i, c in enumerate(chans)]
    xd = [t.to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last).requires_grad_(True) for t in xs]
    xr = [t.to(torch.bfloat16).float().requires_grad_(True) for t in xs]
    y, yr = neck(xd), ref(xr)
    assert tuple(y.shape) == (4, 2048, 4, 4)
    assert rel_err(y.float(), yr) < 2e-2
    w = torch.randn(yr.shape, generator=g)
    (y.float() * w.to(dev)).sum().backward()
    (yr * w).sum().backward()
    assert rel_err(xd[3].grad.float(), xr[3].grad) < 0.12     # bf16 through 4 BatchNorms over 64 samples + ReLU masks
    assert all(t.grad is None for t in xd[:3]) or all(float(t.grad.abs().max()) == 0 for t in xd[:3] if t.grad is not None)
    used = {n for n, p in ref.named_parameters() if p.grad is not None}
    for n, p in neck.named_parameters():
        assert (p.grad is not None) == (n in used), n
    # the dead down-sampling branch still moved its running statistics (training-mode BatchNorm), as in the reference
    assert rel_err(neck.downsamp_modules[0].bn.running_mean, ref.downsamp_modules[0].bn.running_mean) < 2e-2
    bb = T.BACKBONES.get('hrnet_w18')(pretrained=False).to(dev)
    nk = T.NECKS.get('HRNetClassificationNeck')(bb.out_channels).to(dev)
    with torch.no_grad():
        assert tuple(nk(bb(torch.rand(2, 3, 224, 224).to(dev))).shape) == (2, 2048, 7, 7)


@pytest.mark.gpu
def test_branch_streams_are_transparent(monkeypatch):
    """12 SGD steps of the segmentation task: bit-identical parameters with the module branches on their own HIP streams
    (+ weight gradients on the side stream) and with everything on the main stream; the loss falls."""
    from torchok_amd.engine import core as EC
    from torchok_amd.engine import functional as EF
    finals, losses = [], []
    for multi in (True, False):
        monkeypatch.setattr(EC, 'BRANCH_STREAMS', multi)
        monkeypatch.setattr(EF, 'WGRAD_SIDE_STREAM', multi)
        cfg = seg_config('hrnet_w18_small', classes=5, size=64)
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 13)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator().manual_seed(4)
        x = torch.randn(4, 3, 64, 64, generator=g).cuda()
        y = torch.randint(0, 5, (4, 64, 64), generator=g).cuda()
        hist = []
        for it in range(12):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad(set_to_none=True)
            out['loss'].backward()
            opt.step()
            hist.append(out['loss'].detach())
        torch.cuda.synchronize()
        losses.append([float(v) for v in hist])
        finals.append({n: p.detach().clone() for n, p in task.named_parameters()})
    assert losses[0][-1] < losses[0][0]
    assert losses[0] == losses[1]
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n


def test_segmentation_head_hands_the_interpolation_to_its_consumer(dev):
    """SegmentationHead (reference heads/segmentation/base.py:31-41) in training returns the (N, C, H, W) logits as an
    UpsampledLogits: CrossEntropyLoss computes the loss from the low-resolution logits (tok_upsample_ce_*), every other
    consumer gets the interpolated tensor on first touch — same values, same gradients either way; evaluation and the
    one-class head return plain tensors."""
    import torchok_amd as T
    from torchok_amd.losses import cross_entropy as CE
    torch.manual_seed(0)
    head = T.HEADS.get('SegmentationHead')(in_channels=32, num_classes=19).to(dev)
    img = torch.zeros(2, 3, 64, 96, device=dev)
    feat = torch.randn(2, 32, 16, 24, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    tgt = torch.randint(0, 19, (2, 64, 96), generator=torch.Generator().manual_seed(1)).to(dev)
    tgt[:, :3] = 255
    ce = T.LOSSES.get('CrossEntropyLoss')(ignore_index=255)
    res = {}
    for fused in (True, False):
        CE.FUSE_UPSAMPLE_CE = fused
        try:
            head.train()
            head.zero_grad()
            f = feat.clone().requires_grad_(True)
            out = head([img, f])
            assert isinstance(out, CE.UpsampledLogits) == fused
            assert tuple(out.shape) == (2, 19, 64, 96) and out.dim() == 4 and out.dtype == torch.bfloat16
            if fused:
                assert out._full is None                      # metadata queries did not materialise anything
            loss = ce(input=out, target=tgt)
            if fused:
                assert out._full is None                      # ... and neither did the loss
            loss.backward()
            res[fused] = (float(loss.detach()), f.grad.float().cpu(), head.classifier.weight.grad.float().cpu().clone())
        finally:
            CE.FUSE_UPSAMPLE_CE = True
    assert abs(res[True][0] - res[False][0]) < 1e-5 * abs(res[False][0]) + 1e-6
    assert rel_err(res[True][1], res[False][1]) < 2e-3 and rel_err(res[True][2], res[False][2]) < 2e-3
    # any other consumer: the real tensor, with its autograd edge
    f = feat.clone().requires_grad_(True)
    out = head([img, f])
    pred = out.argmax(1)
    assert out._full is not None and tuple(pred.shape) == (2, 64, 96) and not isinstance(pred, CE.UpsampledLogits)
    probs = out.float().softmax(1)
    (probs[:, 0].mean() + ce(input=out, target=tgt)).backward()       # the loss takes the materialised tensor now
    assert f.grad is not None and float(f.grad.float().abs().sum()) > 0
    # evaluation / no_grad: plain tensors
    head.eval()
    with torch.no_grad():
        out = head([img, feat])
    assert type(out) is torch.Tensor and tuple(out.shape) == (2, 19, 64, 96)
    one = T.HEADS.get('SegmentationHead')(in_channels=32, num_classes=1).to(dev).train()
    out1 = one([img, feat.clone().requires_grad_(True)])
    assert type(out1) is torch.Tensor and tuple(out1.shape) == (2, 64, 96)
