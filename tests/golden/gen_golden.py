#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE's own code (runs only where /root/reference
exists, i.e. in the build container; the fixtures it writes are small data files that travel).

What is executed here is the reference itself, imported through a shim:
  * `torchok` is registered as a bare package pointing at /root/reference/torchok so that
    torchok/__init__.py (which imports lightning, albumentations, ...) is NOT executed;
  * the reference's own files are loaded unmodified: constructor/registry.py, constructor/__init__.py,
    models/base.py, models/backbones/base_backbone.py, models/backbones/resnet.py (its ResNet class,
    make_blocks, init_weights, forward), models/poolings/classification/pooling.py,
    models/heads/representation/linear_head.py, models/heads/classification/classification_head.py,
    losses/base.py (JointLoss);
  * the third-party `timm` 0.6.13 (absent offline) is stubbed by oracle/timm_min.py — the
    restated block semantics; this is the part the reference's tests do not pin.
The run also asserts that oracle/torchok_ref.py reproduces the reference wiring bit-for-bit on the
same parameters, i.e. the restatement == reference-files-on-stub-timm.

Parameters are a deterministic function of their state_dict NAME (tests/helpers.py:
deterministic_state), so any box can rebuild the exact same model without the reference.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
REF = '/root/reference/torchok'

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from oracle import timm_min  # noqa: E402
import oracle.torchok_ref as R  # noqa: E402
from helpers import deterministic_state  # noqa: E402


def _fake_pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    sys.modules[name] = m
    return m


def _load(name, file, pkg_dir=None):
    spec = importlib.util.spec_from_file_location(name, file, submodule_search_locations=[pkg_dir] if pkg_dir else None)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_shim():
    # ---- timm stub (restated subset) -------------------------------------------------------
    import re
    _fake_pkg('timm')
    d = _fake_pkg('timm.data')
    d.IMAGENET_DEFAULT_MEAN, d.IMAGENET_DEFAULT_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    _fake_pkg('timm.models')
    h = _fake_pkg('timm.models.helpers')
    h.build_model_with_cfg = timm_min.build_model_with_cfg
    lay = _fake_pkg('timm.models.layers')
    lay.BlurPool2d, lay.DropPath, lay.get_attn, lay.GroupNorm = None, timm_min.DropPath, (lambda *a, **k: None), nn.GroupNorm
    ap = _fake_pkg('timm.models.layers.adaptive_avgmax_pool')
    ap.SelectAdaptivePool2d = timm_min.SelectAdaptivePool2d
    rn = _fake_pkg('timm.models.resnet')
    for n in ('BasicBlock', 'Bottleneck', 'create_aa', 'drop_blocks', 'downsample_avg', 'downsample_conv'):
        setattr(rn, n, getattr(timm_min, n))
    ft = _fake_pkg('timm.models.features')
    ft.FeatureHooks = timm_min.FeatureHooks
    rg = _fake_pkg('timm.models.registry')
    rg._natural_key = lambda s: [int(p) if p.isdigit() else p for p in re.split(r'(\d+)', s.lower())]
    # ---- reference files -------------------------------------------------------------------------
    _fake_pkg('torchok', REF)
    _load('torchok.constructor', f'{REF}/constructor/__init__.py', f'{REF}/constructor')
    _fake_pkg('torchok.models', f'{REF}/models')
    _load('torchok.models.base', f'{REF}/models/base.py')
    bb = _fake_pkg('torchok.models.backbones', f'{REF}/models/backbones')
    bb.BaseBackbone = _load('torchok.models.backbones.base_backbone', f'{REF}/models/backbones/base_backbone.py').BaseBackbone
    resnet = _load('torchok.models.backbones.resnet', f'{REF}/models/backbones/resnet.py')
    _fake_pkg('torchok.models.poolings', f'{REF}/models/poolings')
    _fake_pkg('torchok.models.poolings.classification', f'{REF}/models/poolings/classification')
    pooling = _load('torchok.models.poolings.classification.pooling', f'{REF}/models/poolings/classification/pooling.py')
    _fake_pkg('torchok.models.heads', f'{REF}/models/heads')
    _fake_pkg('torchok.models.heads.representation', f'{REF}/models/heads/representation')
    _load('torchok.models.heads.representation.linear_head', f'{REF}/models/heads/representation/linear_head.py')
    _fake_pkg('torchok.models.heads.classification', f'{REF}/models/heads/classification')
    head = _load('torchok.models.heads.classification.classification_head',
                 f'{REF}/models/heads/classification/classification_head.py')
    _fake_pkg('torchok.losses', f'{REF}/losses')
    losses = _load('torchok.losses.base', f'{REF}/losses/base.py')
    return resnet, pooling, head, losses


def golden_hrnet(out_path, variant='hrnet_w18_small', classes=19, batch=4, size=128, seed=41):
    """SegmentationTask wiring (tasks/segmentation.py:60-93) over the reference's OWN hrnet.py, segmentation neck
    and head (timm.models.hrnet stubbed by oracle/hrnet_ref.py); asserts the restatement is bit-identical."""
    import oracle.hrnet_ref as H
    hr = _fake_pkg('timm.models.hrnet')
    for n in ('_BN_MOMENTUM', 'BasicBlock', 'blocks_dict', 'Bottleneck', 'cfg_cls', 'HighResolutionModule'):
        setattr(hr, n, getattr(H, n))
    hrnet = _load('torchok.models.backbones.hrnet', f'{REF}/models/backbones/hrnet.py')
    _fake_pkg('torchok.models.modules', f'{REF}/models/modules')
    _fake_pkg('torchok.models.modules.bricks', f'{REF}/models/modules/bricks')
    _load('torchok.models.modules.bricks.convbnact', f'{REF}/models/modules/bricks/convbnact.py')
    _fake_pkg('torchok.models.necks', f'{REF}/models/necks')
    _fake_pkg('torchok.models.necks.segmentation', f'{REF}/models/necks/segmentation')
    neck = _load('torchok.models.necks.segmentation.hrnet', f'{REF}/models/necks/segmentation/hrnet.py')
    _fake_pkg('torchok.models.heads.segmentation', f'{REF}/models/heads/segmentation')
    head = _load('torchok.models.heads.segmentation.base', f'{REF}/models/heads/segmentation/base.py')

    class RefSeg(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = getattr(hrnet, variant)(pretrained=False, in_channels=3)
            self.neck = neck.HRNetSegmentationNeck(in_channels=self.backbone.out_encoder_channels)
            self.head = head.SegmentationHead(in_channels=self.neck.out_channels, num_classes=classes)

        def forward_with_gt(self, batch):
            feats = self.backbone.forward_features(batch['image'])
            return {'prediction': self.head(self.neck(feats)), 'target': batch['target']}

    torch.manual_seed(seed)
    task = RefSeg().train()
    task.load_state_dict(deterministic_state(task.state_dict(), seed))
    ora = H.SegmentationModel(variant, classes).train()
    assert set(ora.state_dict()) == set(task.state_dict()), 'state_dict keys differ: restated wiring != reference'
    ora.load_state_dict(deterministic_state(ora.state_dict(), seed))

    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g).half().float()
    y = torch.randint(0, classes, (batch, size, size), generator=g)
    y[:, :4, :] = 255                                    # ignored border, as segmentation datasets have
    ce = nn.CrossEntropyLoss(ignore_index=255)
    feats = task.backbone.forward_features(x)
    out = task.forward_with_gt({'image': x, 'target': y})
    loss = ce(out['prediction'], y)
    loss.backward()
    oout = ora.forward_with_gt({'image': x, 'target': y})
    oloss = ce(oout['prediction'], y)
    oloss.backward()
    assert torch.equal(out['prediction'], oout['prediction']) and torch.equal(loss, oloss)
    for (n1, p1), (n2, p2) in zip(task.named_parameters(), ora.named_parameters()):
        assert n1 == n2 and torch.equal(p1.grad, p2.grad), n1
    names = [n for n, _ in task.named_parameters()]
    grads = {n: p.grad for n, p in task.named_parameters()}
    small = [n for n in names if grads[n].numel() <= 1024][:60]
    np.savez_compressed(
        out_path, variant=variant, num_classes=classes, seed=seed, x=x.half().numpy(), y=y.numpy().astype(np.uint8),
        feat_shapes=np.array([list(f.shape) for f in feats[1:]]),
        feat_sumsq=np.array([float((f.double() ** 2).sum()) for f in feats[1:]]),
        prediction=out['prediction'].detach().half().numpy(), loss=float(loss.detach()),
        param_names=np.array(names), grad_norm=np.array([float(grads[n].double().norm()) for n in names]),
        small_names=np.array(small), **{f'grad__{n}': grads[n].numpy() for n in small})
    print(f'wrote {out_path}: loss {float(loss.detach()):.6f}, {len(names)} params, restatement == reference files: OK')


def golden_swin(out_path, seed=51):
    """ClassificationTask wiring (tasks/classification.py:90-119) over the reference's OWN swin.py
    (SwinTransformerV2 / BasicLayer override), Pooling and ClassificationHead; timm.models.swin_transformer_v2 and
    timm.models.layers are stubbed by oracle/swin_ref.py.  Asserts the restated wiring (oracle SwinV2) is
    bit-identical to the reference's class."""
    import oracle.swin_ref as S
    lay = sys.modules['timm.models.layers']
    lay.trunc_normal_, lay.to_2tuple = S.trunc_normal_, S.to_2tuple
    sw = _fake_pkg('timm.models.swin_transformer_v2')
    for n in ('BasicLayer', 'checkpoint_filter_fn', 'PatchEmbed', 'PatchMerging'):
        setattr(sw, n, getattr(S, n))
    # timm's build_model_with_cfg forwards unknown kwargs (pretrained_filter_fn) to its loader, not the model
    h = sys.modules['timm.models.helpers']
    base_build = h.build_model_with_cfg
    h.build_model_with_cfg = lambda cls, variant, pretrained, pretrained_filter_fn=None, **kw: \
        base_build(cls, variant, pretrained, **kw)
    swin = _load('torchok.models.backbones.swin', f'{REF}/models/backbones/swin.py')
    pooling = sys.modules['torchok.models.poolings.classification.pooling']
    head = sys.modules['torchok.models.heads.classification.classification_head']
    kw = dict(img_size=64, window_size=4, depths=(2, 2, 2, 2), drop_path_rate=0.0)
    classes, batch = 10, 4

    class RefCls(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = swin.swinv2_custom(pretrained=False, in_channels=3, **kw)
            self.pooling = pooling.Pooling(in_channels=self.backbone.out_channels)
            self.head = head.ClassificationHead(in_channels=self.pooling.out_channels, num_classes=classes)

    torch.manual_seed(seed)
    task = RefCls().train()
    sd = deterministic_state(task.state_dict(), seed)
    task.load_state_dict(sd)
    ora = S.SwinV2(**kw).train()
    bsd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    assert set(ora.state_dict()) == set(bsd), 'state_dict keys differ: restated wiring != reference'
    ora.load_state_dict(bsd)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, 64, 64, generator=g).half().float()
    y = torch.randint(0, classes, (batch,), generator=g)
    feats = task.backbone.forward_features(x)
    ofeats = ora.forward_features(x)
    assert all(torch.equal(a, b) for a, b in zip(feats, ofeats))
    assert torch.equal(task.backbone(x), ora(x))
    emb = task.pooling(task.backbone(x))
    pred = task.head(emb, y)
    loss = nn.functional.cross_entropy(pred, y)
    loss.backward()
    opt = torch.optim.AdamW(task.parameters(), lr=1e-3, weight_decay=0.05)
    # feature_norms.0-2 take no part in `backbone(x)` (swin.py:251-256): no gradient, exactly as in the reference
    grads = {n: p.grad.clone() for n, p in task.named_parameters() if p.grad is not None}
    opt.step()
    names = [n for n, _ in task.named_parameters() if n in grads]
    no_grad = [n for n, _ in task.named_parameters() if n not in grads]
    small = [n for n in names if grads[n].numel() <= 768][:80]
    np.savez_compressed(
        out_path, seed=seed, num_classes=classes, x=x.half().numpy(), y=y.numpy(),
        feat_shapes=np.array([list(f.shape) for f in feats[1:]]),
        feat_sumsq=np.array([float((f.double() ** 2).sum()) for f in feats[1:]]),
        last_feature=feats[-1].detach().numpy(), prediction=pred.detach().numpy(), loss=float(loss.detach()),
        param_names=np.array(names), no_grad_names=np.array(no_grad),
        grad_norm=np.array([float(grads[n].double().norm()) for n in names]),
        post_step_norm=np.array([float(task.get_parameter(n).double().norm()) for n in names]),
        small_names=np.array(small), **{f'grad__{n}': grads[n].numpy() for n in small})
    print(f'wrote {out_path}: loss {float(loss.detach()):.6f}, {len(names)} params, restatement == reference files: OK')


def golden_davit(out_path, seed=71):
    """ClassificationTask wiring over the reference's OWN davit.py — its only in-tree transformer: every class in that
    file is reference code, the stubs cover just `timm.models.layers.{DropPath, trunc_normal_, to_2tuple}` and
    `build_model_with_cfg`.  drop_path_rate > 0 in training mode: the stochastic-depth draws are part of the step.
    Asserts oracle/davit_ref.py is bit-identical (features, logits, gradients)."""
    import oracle.davit_ref as D
    import oracle.swin_ref as S
    lay = sys.modules['timm.models.layers']
    lay.trunc_normal_, lay.to_2tuple, lay.DropPath = S.trunc_normal_, S.to_2tuple, timm_min.DropPath
    if 'torchok.models.modules' not in sys.modules:
        _fake_pkg('torchok.models.modules', f'{REF}/models/modules')
        _fake_pkg('torchok.models.modules.bricks', f'{REF}/models/modules/bricks')
    _load('torchok.models.modules.bricks.mlp', f'{REF}/models/modules/bricks/mlp.py')
    # timm's build_model_with_cfg keeps pretrained_cfg / pretrained_filter_fn for its checkpoint loader
    h = sys.modules['timm.models.helpers']
    h.build_model_with_cfg = lambda cls, variant, pretrained, pretrained_cfg=None, pretrained_filter_fn=None, **kw_: \
        timm_min.build_model_with_cfg(cls, variant, pretrained, **kw_)
    davit = _load('torchok.models.backbones.davit', f'{REF}/models/backbones/davit.py')
    pooling = sys.modules['torchok.models.poolings.classification.pooling']
    head = sys.modules['torchok.models.heads.classification.classification_head']
    kw = dict(img_size=128, window_size=4, drop_path_rate=0.2)
    classes, batch = 10, 4

    class RefCls(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = davit.davit_t(pretrained=False, in_channels=3, **kw)
            self.pooling = pooling.Pooling(in_channels=self.backbone.out_channels)
            self.head = head.ClassificationHead(in_channels=self.pooling.out_channels, num_classes=classes)

    torch.manual_seed(seed)
    task = RefCls().train()
    sd = deterministic_state(task.state_dict(), seed)
    task.load_state_dict(sd)
    ora = D.davit_t(window_size=4, drop_path_rate=0.2).train()
    bsd = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    assert set(ora.state_dict()) == set(bsd), 'state_dict keys differ: restated wiring != reference'
    ora.load_state_dict(bsd)
    # cpe_act=True (ConvPosEnc with its GELU residual): restatement == reference on features and gradients, too
    rc = davit.davit_t(pretrained=False, in_channels=3, img_size=128, window_size=4, drop_path_rate=0.0, cpe_act=True).train()
    oc = D.davit_t(window_size=4, drop_path_rate=0.0, cpe_act=True).train()
    csd = deterministic_state(rc.state_dict(), seed + 5)
    rc.load_state_dict(csd)
    oc.load_state_dict(csd)
    xc = torch.randn(2, 3, 128, 128, generator=torch.Generator().manual_seed(seed + 6))
    fr, fo = rc.forward_features(xc), oc.forward_features(xc)
    assert all(torch.equal(a, b) for a, b in zip(fr, fo))
    sum(f.sum() for f in fr[1:]).backward()
    sum(f.sum() for f in fo[1:]).backward()
    for (n, p), (n2, p2) in zip(rc.named_parameters(), oc.named_parameters()):
        assert n == n2 and p.grad is not None and torch.equal(p.grad, p2.grad), n
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, 128, 128, generator=g).half().float()
    y = torch.randint(0, classes, (batch,), generator=g)
    task.eval(), ora.eval()
    feats = task.backbone.forward_features(x)
    ofeats = ora.forward_features(x)
    assert all(torch.equal(a, b) for a, b in zip(feats, ofeats))
    task.train(), ora.train()
    torch.manual_seed(seed + 2)
    state = torch.get_rng_state()
    last = task.backbone(x)
    pred = task.head(task.pooling(last), y)
    loss = nn.functional.cross_entropy(pred, y)
    loss.backward()
    torch.set_rng_state(state)
    olast = ora(x)
    assert torch.equal(last, olast)
    nn.functional.cross_entropy(task.head(task.pooling(olast), y), y).backward()
    # the per-sample stochastic-depth factors, in draw order (2 per block): what a device RNG cannot reproduce
    torch.set_rng_state(state)
    dpr = [m.drop_prob for m in task.backbone.modules() if type(m).__name__ == 'DropPath' for _ in range(2)]
    draws = np.stack([(torch.empty(batch, 1, 1).bernoulli_(1 - p) / (1 - p)).reshape(-1).numpy() if p > 0
                      else np.ones(batch, np.float32) for p in dpr])
    grads = {n: p.grad.clone() for n, p in task.named_parameters() if p.grad is not None}
    for n, p in ora.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, grads['backbone.' + n]), n
    opt = torch.optim.AdamW(task.parameters(), lr=1e-3, weight_decay=0.05)
    opt.step()
    names = [n for n, _ in task.named_parameters() if n in grads]
    no_grad = [n for n, _ in task.named_parameters() if n not in grads]
    small = [n for n in names if grads[n].numel() <= 768][:80]
    np.savez_compressed(
        out_path, seed=seed, num_classes=classes, x=x.half().numpy(), y=y.numpy(), drop_scales=draws,
        feat_shapes=np.array([list(f.shape) for f in feats[1:]]),
        feat_sumsq=np.array([float((f.double() ** 2).sum()) for f in feats[1:]]),
        eval_last_feature=feats[-1].detach().numpy(),
        last_feature=last.detach().numpy(), prediction=pred.detach().numpy(), loss=float(loss.detach()),
        param_names=np.array(names), no_grad_names=np.array(no_grad),
        grad_norm=np.array([float(grads[n].double().norm()) for n in names]),
        post_step_norm=np.array([float(task.get_parameter(n).double().norm()) for n in names]),
        small_names=np.array(small), **{f'grad__{n}': grads[n].numpy() for n in small})
    print(f'wrote {out_path}: loss {float(loss.detach()):.6f}, {len(names)} params with gradients, {len(no_grad)} without, '
          f'restatement == reference files: OK')


def golden_ocr(out_path, seed=81):
    """OCRSegmentationHead: the reference's OWN heads/segmentation/ocr.py and modules/bricks/convbnact.py (in-tree files, torch
    only) in training mode — both outputs, the channel-dropout factors it drew, every gradient of an (out, out_aux) loss;
    asserts oracle/ocr_ref.py is bit-identical."""
    import oracle.ocr_ref as O
    if 'torchok.models.modules' not in sys.modules:
        _fake_pkg('torchok.models.modules', f'{REF}/models/modules')
        _fake_pkg('torchok.models.modules.bricks', f'{REF}/models/modules/bricks')
    if 'torchok.models.modules.bricks.convbnact' not in sys.modules:
        _load('torchok.models.modules.bricks.convbnact', f'{REF}/models/modules/bricks/convbnact.py')
    if 'torchok.models.heads.segmentation' not in sys.modules:
        _fake_pkg('torchok.models.heads.segmentation', f'{REF}/models/heads/segmentation')
    ocr = _load('torchok.models.heads.segmentation.ocr', f'{REF}/models/heads/segmentation/ocr.py')
    cin, classes, batch, h, w = 48, 7, 4, 16, 24
    torch.manual_seed(seed)
    ref = ocr.OCRSegmentationHead(in_channels=cin, num_classes=classes).train()
    ora = O.OCRSegmentationHead(cin, classes).train()
    assert set(ref.state_dict()) == set(ora.state_dict()), 'state_dict keys differ: restated wiring != reference'
    sd = deterministic_state(ref.state_dict(), seed)
    ref.load_state_dict(sd)
    ora.load_state_dict(sd)
    g = torch.Generator().manual_seed(seed + 1)
    image = torch.zeros(batch, 3, 4 * h, 4 * w)
    feats = torch.randn(batch, cin, h, w, generator=g).half().float()
    wo, wa = torch.randn(batch, classes, 4 * h, 4 * w, generator=g), torch.randn(batch, classes, 4 * h, 4 * w, generator=g)
    outs = []
    for m in (ref, ora):
        torch.manual_seed(seed + 2)
        x = feats.clone().requires_grad_(True)
        out, aux = m([image, x])
        ((out * wo).sum() + 0.4 * (aux * wa).sum()).backward()
        outs.append((out.detach(), aux.detach(), x.grad.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    for (n, p), (n2, p2) in zip(ref.named_parameters(), ora.named_parameters()):
        assert n == n2 and torch.equal(p.grad, p2.grad), n
    torch.manual_seed(seed + 2)
    drop = torch.empty(batch, 128, 1, 1).bernoulli_(0.95).div_(0.95).reshape(batch, 128)
    ref.eval()
    with torch.no_grad():
        ev = ref([image, feats])
    names = [n for n, _ in ref.named_parameters()]
    np.savez_compressed(
        out_path, seed=seed, in_channels=cin, num_classes=classes, feats=feats.numpy(), image_hw=np.array([4 * h, 4 * w]),
        weight_seed=seed + 1,        # w_out / w_aux: the two randn draws that follow `feats` on torch.Generator(weight_seed)
        drop_scale=drop.numpy(), out=outs[0][0].half().numpy(), out_aux=outs[0][1].half().numpy(),
        d_feats=outs[0][2].numpy(), eval_out=ev.half().numpy(), param_names=np.array(names),
        grad_norm=np.array([float(p.grad.double().norm()) for _, p in ref.named_parameters()]),
        **{f'grad__{n}': p.grad.numpy() for n, p in ref.named_parameters() if p.numel() <= 4096})
    print(f'wrote {out_path}: {len(names)} params, restatement == reference files: OK')


def golden_dice(out_path):
    """DiceLoss outputs and input gradients from the reference's own losses/segmentation/dice.py (imports as is)."""
    _fake_pkg('torchok.losses.segmentation', f'{REF}/losses/segmentation')
    dice = _load('torchok.losses.segmentation.dice', f'{REF}/losses/segmentation/dice.py')
    g = torch.Generator().manual_seed(61)
    out = {}
    z = torch.randn(3, 5, 12, 10, generator=g) * 2
    t = torch.randint(0, 4, (3, 12, 10), generator=g)          # class 4 never occurs: masked (dice.py:180-181)
    for tag, kw in (('mc', {}), ('mc_log', dict(log_loss=True, smooth=1.0)), ('mc_sel', dict(classes=torch.tensor([0, 2, 4])))):   # a LIST goes through np.ndarray(x) in the reference
                                                                      # (dice.py:74: uninitialised array of that SHAPE) -> NaN
        zi = z.clone().bfloat16().float().requires_grad_(True)
        L = dice.DiceLoss('multiclass', **kw)(zi, t)
        L.backward()
        out[tag + '_loss'], out[tag + '_dz'] = float(L.detach()), zi.grad.numpy()
    zb = torch.randn(3, 12, 10, generator=g) * 2
    tb = (torch.rand(3, 12, 10, generator=g) < 0.3).float()
    zi = zb.clone().bfloat16().float().requires_grad_(True)
    L = dice.DiceLoss('binary', smooth=0.5)(zi, tb)
    L.backward()
    out.update(bin_loss=float(L.detach()), bin_dz=zi.grad.numpy(), z=z.numpy(), t=t.numpy(), zb=zb.numpy(), tb=tb.numpy())
    tm = (torch.rand(3, 5, 12, 10, generator=g) < 0.3).float()       # multilabel targets on the logits z; class 3 stays empty
    tm[:, 3] = 0
    for tag, kw in (('ml', dict(smooth=0.5)), ('ml_log_sel', dict(log_loss=True, smooth=1.0, classes=torch.tensor([1, 3, 4])))):
        zi = z.clone().bfloat16().float().requires_grad_(True)
        L = dice.DiceLoss('multilabel', **kw)(zi, tm)
        L.backward()
        out[tag + '_loss'], out[tag + '_dz'] = float(L.detach()), zi.grad.numpy()
    out['tm'] = tm.numpy()
    np.savez_compressed(out_path, **out)
    print('wrote', out_path)


def golden_bce(out_path):
    """BCEWithLogitsLoss (ignore value) outputs and input gradients from the reference's own
    losses/classification/binary_cross_entropy.py (imports as is: torch + the registry)."""
    _fake_pkg('torchok.losses.classification', f'{REF}/losses/classification')
    mod = _load('torchok.losses.classification.binary_cross_entropy', f'{REF}/losses/classification/binary_cross_entropy.py')
    g = torch.Generator().manual_seed(91)
    out = {}
    x = (torch.randn(37, 21, generator=g) * 3).bfloat16().float()          # 21 classes: row pitch 24 on the device
    t = (torch.rand(37, 21, generator=g) < 0.3).float()
    t[torch.rand(37, 21, generator=g) < 0.25] = -1                         # ignored labels
    x4 = (torch.randn(2, 5, 6, 8, generator=g) * 2).bfloat16().float()     # any shape goes through the mask
    t4 = (torch.rand(2, 5, 6, 8, generator=g) < 0.5).float()
    t4[torch.rand(2, 5, 6, 8, generator=g) < 0.1] = -1
    for tag, (xi, ti) in (('a', (x, t)), ('b', (x4, t4))):
        for red in ('mean', 'sum'):
            xv = xi.clone().requires_grad_(True)
            L = mod.BCEWithLogitsLoss(reduction=red)(xv, ti)
            L.backward()
            out[f'{tag}_{red}_loss'], out[f'{tag}_{red}_dx'] = float(L.detach()), xv.grad.numpy()
    xv = x.clone().requires_grad_(True)
    L = mod.BCEWithLogitsLoss(ignore_index=0)(xv, t.clamp_min(0))          # ignoring the negatives: only t == 1 counts
    L.backward()
    out['ign0_loss'], out['ign0_dx'] = float(L.detach()), xv.grad.numpy()
    L = mod.BCEWithLogitsLoss()(x, torch.full_like(t, -1))                 # nothing selected -> 0 (:58-59)
    out['empty_loss'] = float(L)
    out.update(x=x.numpy(), t=t.numpy(), x4=x4.numpy(), t4=t4.numpy())
    np.savez_compressed(out_path, **out)
    print('wrote', out_path)


def golden_unsupervised(out_path):
    """NT_XentLoss from the reference's own losses/representation/unsupervised.py; TripletMarginLoss is torch's class
    (what the reference registers, losses/__init__.py:39)."""
    if 'torchok.losses.representation' not in sys.modules:
        _fake_pkg('torchok.losses.representation', f'{REF}/losses/representation')
    un = _load('torchok.losses.representation.unsupervised', f'{REF}/losses/representation/unsupervised.py')
    g = torch.Generator().manual_seed(71)
    out = {}
    e1 = torch.nn.functional.normalize(torch.randn(12, 40, generator=g)).bfloat16().float().requires_grad_(True)
    e2 = torch.nn.functional.normalize(torch.randn(12, 40, generator=g)).bfloat16().float().requires_grad_(True)
    L = un.NT_XentLoss(temperature=0.2)(e1, e2)
    L.backward()
    out.update(ntx_e1=e1.detach().numpy(), ntx_e2=e2.detach().numpy(), ntx_loss=float(L.detach()), ntx_d1=e1.grad.numpy(),
               ntx_d2=e2.grad.numpy())
    a, p, n = ((torch.randn(16, 24, generator=g) * 0.7).bfloat16().float().requires_grad_(True) for _ in range(3))
    for tag, kw in (('tri', dict(margin=1.0)), ('tri_swap', dict(margin=0.5, swap=True))):
        for t in (a, p, n):
            t.grad = None
        L = torch.nn.TripletMarginLoss(**kw)(a, p, n)
        L.backward()
        out.update({f'{tag}_loss': float(L.detach()), f'{tag}_da': a.grad.numpy().copy(), f'{tag}_dp': p.grad.numpy().copy(),
                    f'{tag}_dn': n.grad.numpy().copy()})
    out.update(tri_a=a.detach().numpy(), tri_p=p.detach().numpy(), tri_n=n.detach().numpy())
    np.savez_compressed(out_path, **out)
    print('wrote', out_path)


def golden_metric(out_path):
    """ArcFaceHead / LinearHead(normalize) / ContrastiveLoss / calc_relevance_matrix from the reference's
    own files; asserts oracle/metric_ref.py == reference bit-for-bit on the same inputs."""
    import ast
    import oracle.metric_ref as M
    arc = _load('torchok.models.heads.classification.arcface_head',
                f'{REF}/models/heads/classification/arcface_head.py')
    lin = sys.modules['torchok.models.heads.representation.linear_head']
    _fake_pkg('torchok.losses.representation', f'{REF}/losses/representation')
    pw = _load('torchok.losses.representation.pairwise', f'{REF}/losses/representation/pairwise.py')
    # calc_relevance_matrix: tasks/pairwise_task.py cannot be imported (its base class pulls in lightning), so
    # the method's own source lines are compiled out of the reference file and bound to a stub `self`
    src = open(f'{REF}/tasks/pairwise_task.py').read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == 'calc_relevance_matrix')
    ns = {'torch': torch, 'Tensor': torch.Tensor}
    exec(compile(ast.Module([fn], []), f'{REF}/tasks/pairwise_task.py', 'exec'), ns)
    calc_rel = ns['calc_relevance_matrix']

    g = torch.Generator().manual_seed(21)
    out = {}
    # ---- ArcFaceHead: default + easy margin, train + eval, gradients of CE(output, target) -------
    n, c, k = 16, 64, 10
    x = torch.randn(n, c, generator=g)
    t = torch.randint(0, k, (n,), generator=g)
    for tag, easy in (('arc', False), ('arc_easy', True)):
        h = arc.ArcFaceHead(c, k, easy_margin=easy).train()
        h.load_state_dict(deterministic_state(h.state_dict(), 21))
        with torch.no_grad():
            h.weight.mul_(4.0)        # rows far from unit norm: exercises the weight normalisation backward
        xi = x.clone().requires_grad_(True)
        y = h(xi, t)
        F_ce = torch.nn.functional.cross_entropy(y, t)
        F_ce.backward()
        yo = M.arcface_forward(x, h.weight.detach(), t, h.margin, h.scale, easy)
        assert torch.equal(y.detach(), yo), 'arcface restatement != reference'
        sc, mg = M.arcface_defaults(c, k)
        assert sc == h.scale and mg == h.margin
        out.update({f'{tag}_out': y.detach().numpy(), f'{tag}_dx': xi.grad.numpy(), f'{tag}_dw': h.weight.grad.numpy(),
                    f'{tag}_loss': float(F_ce), f'{tag}_eval': h.eval()(x).detach().numpy(),
                    f'{tag}_scale': h.scale, f'{tag}_margin': h.margin})
    out.update(arc_x=x.numpy(), arc_t=t.numpy(), arc_w=h.weight.detach().numpy())
    # ---- LinearHead(normalize=True) -----------------------------------------------------------------
    lh = lin.LinearHead(c, 24, normalize=True)
    lh.load_state_dict(deterministic_state(lh.state_dict(), 22))
    xi = x.clone().requires_grad_(True)
    y = lh(xi)
    (y * torch.linspace(-1, 1, 24)).sum().backward()
    assert torch.equal(y.detach(), M.linear_head_forward(x, lh.fc.weight.detach(), lh.fc.bias.detach(), True))
    out.update(lin_out=y.detach().numpy(), lin_dx=xi.grad.numpy(), lin_dw=lh.fc.weight.grad.numpy(),
               lin_db=lh.fc.bias.grad.numpy())
    # ---- relevance matrix + ContrastiveLoss (emb1 is emb2, as PairwiseLearnTask hands them over) -------
    class _Self:
        num_classes = 6
    lab = torch.randint(0, 6, (n,), generator=g)
    R = calc_rel(_Self(), lab)
    assert torch.equal(R, M.relevance_matrix(lab, 6))
    e = (0.6 * torch.randn(n, 24, generator=g)).requires_grad_(True)
    cl = pw.ContrastiveLoss(margin=1.0)
    L = cl(emb1=e, emb2=e, R=R)
    L.backward()
    assert torch.equal(L.detach(), M.contrastive_loss(e.detach(), e.detach(), R, 1.0))
    e2 = (0.6 * torch.randn(12, 24, generator=g)).requires_grad_(True)
    e1 = e.detach().clone().requires_grad_(True)
    lab2 = torch.randint(0, 6, (12,), generator=g)
    R2 = (lab[:, None] == lab2[None, :]).float()
    L2 = cl(emb1=e1, emb2=e2, R=R2)
    L2.backward()
    ml = (torch.rand(n, 9, generator=torch.Generator().manual_seed(77)) < 0.2).float()     # multi-label rows, some empty
    Rml = calc_rel(_Self(), ml)
    assert torch.equal(Rml, M.relevance_matrix(ml, 9))
    out.update(con_ml=ml.numpy(), con_Rml=Rml.numpy())
    for tag, kw in (('l1', dict(reg='L1')), ('l2', dict(reg='L2', eps=0.05)), ('sum', dict(reduction='sum')),
                    ('l1sum', dict(reg='L1', reduction='sum', eps=0.01))):
        ei = e.detach().clone().requires_grad_(True)
        Lr = pw.ContrastiveLoss(margin=1.0, **kw)(emb1=ei, emb2=ei, R=R)
        Lr.backward()
        assert torch.equal(Lr.detach(), M.contrastive_loss(e.detach(), e.detach(), R, 1.0, **kw))
        out[f'con_{tag}_loss'], out[f'con_{tag}_de'] = float(Lr), ei.grad.numpy()
    out.update(con_lab=lab.numpy(), con_R=R.numpy(), con_e=e.detach().numpy(), con_loss=float(L), con_de=e.grad.numpy(),
               con_e2=e2.detach().numpy(), con_R2=R2.numpy(), con_loss2=float(L2), con_de1=e1.grad.numpy(),
               con_de2=e2.grad.numpy())
    np.savez_compressed(out_path, **out)
    print(f'wrote {out_path}: restatement == reference files: OK')


class RefTask(nn.Module):
    """The wiring of reference tasks/classification.py:45-73,90-119 over the reference's own modules."""

    def __init__(self, mods, backbone, num_classes):
        super().__init__()
        resnet, pooling, head, _ = mods
        self.backbone = getattr(resnet, backbone)(pretrained=False, in_channels=3)
        self.pooling = pooling.Pooling(in_channels=self.backbone.out_channels)
        self.head = head.ClassificationHead(in_channels=self.pooling.out_channels, num_classes=num_classes)

    def forward_with_gt(self, batch):
        features = self.backbone(batch['image'])
        embeddings = self.pooling(features)
        prediction = self.head(embeddings, batch['target'])
        return {'embeddings': embeddings, 'prediction': prediction, 'target': batch['target']}


def golden_step(mods, backbone, num_classes, batch, size, seed, out_path):
    torch.manual_seed(seed)
    task = RefTask(mods, backbone, num_classes).train()
    task.load_state_dict(deterministic_state(task.state_dict(), seed))
    ora = R.ClassificationModel(backbone, num_classes).train()
    ora.load_state_dict(deterministic_state(ora.state_dict(), seed))
    assert set(ora.state_dict()) == set(task.state_dict()), 'state_dict keys differ: restated wiring != reference'

    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, 3, size, size, generator=g).half().float()   # stored as fp16 (exact), small fixture
    y = torch.randint(0, num_classes, (batch,), generator=g)
    JointLoss = mods[3].JointLoss
    jl = JointLoss([nn.CrossEntropyLoss()], [dict(input='prediction', target='target')], [None], [None])
    opt = torch.optim.SGD(task.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)

    feats = task.backbone.forward_features(x)
    out = task.forward_with_gt({'image': x, 'target': y})
    total, tagged = jl(**out)
    total.backward()
    grads = {n: p.grad.clone() for n, p in task.named_parameters()}
    opt.step()

    # the restatement must agree with the reference files bit-for-bit (same ATen ops, same order)
    ofeats = ora.backbone.forward_features(x)
    oloss, oout = R.training_step(ora, {'image': x, 'target': y},
                                  torch.optim.SGD(ora.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4))
    assert all(torch.equal(a, b) for a, b in zip(feats, ofeats))
    assert torch.equal(out['prediction'], oout['prediction']) and torch.equal(total.detach(), oloss)
    for (n, p), (n2, p2) in zip(task.named_parameters(), ora.named_parameters()):
        assert n == n2 and torch.equal(p, p2), n
    for (n, b), (n2, b2) in zip(task.named_buffers(), ora.named_buffers()):
        assert n == n2 and torch.equal(b, b2), n

    names = [n for n, _ in task.named_parameters()]
    small = [n for n in names if task.get_parameter(n).numel() <= 2048]
    np.savez_compressed(
        out_path,
        backbone=backbone, num_classes=num_classes, seed=seed,
        x=x.half().numpy(), y=y.numpy(),
        feat_shapes=np.array([list(f.shape) for f in feats]),
        feat_sum=np.array([float(f.double().sum()) for f in feats]),
        feat_sumsq=np.array([float((f.double() ** 2).sum()) for f in feats]),
        embeddings=out['embeddings'].detach().numpy(), prediction=out['prediction'].detach().numpy(),
        loss=float(total),
        param_names=np.array(names),
        grad_norm=np.array([float(grads[n].double().norm()) for n in names]),
        post_step_norm=np.array([float(task.get_parameter(n).double().norm()) for n in names]),
        small_names=np.array(small),
        **{f'grad__{n}': grads[n].numpy() for n in small},
        **{f'post__{n}': task.get_parameter(n).detach().numpy() for n in small},
        bn1_running_mean=task.backbone.bn1.running_mean.numpy(), bn1_running_var=task.backbone.bn1.running_var.numpy(),
        bn1_nbt=int(task.backbone.bn1.num_batches_tracked),
    )
    print(f'wrote {out_path}: loss {float(total):.6f}, {len(names)} params, restatement == reference files: OK')


def golden_heads(mods, out_path):
    _, _, head, losses = mods
    torch.manual_seed(3)
    h = head.ClassificationHead(in_channels=32, num_classes=7)
    h.load_state_dict(deterministic_state(h.state_dict(), 3))
    x = torch.randn(5, 32, generator=torch.Generator().manual_seed(4))
    np.savez_compressed(out_path, x=x.numpy(), y=h(x).detach().numpy(),
                        binary=head.ClassificationHead(in_channels=32, num_classes=1)(x).shape)
    print('wrote', out_path)


def golden_retrieval(out_known, out_meters):
    """Retrieval meters.  (a) The reference's own known-answer tables
    (tests/base_tests/metrics/representation/data.py) are copied as DATA.  (b) The reference's own
    metrics/index_base_metric.py + representation_ranx.py run unmodified on random data with the absent third-party
    packages stubbed: faiss.IndexFlat{IP,L2} -> exhaustive search (oracle.retrieval_ref.flat_search), ranx.metrics ->
    the restated per-query functions, torchmetrics.Metric -> a bare state holder.  One adapter: the list returned by
    RanxBasedMeter.process_data_for_metric_func gets a `.shape`, which index_base_metric.py:262 reads (a plain list
    has none, so compute() of the shipped file raises there).  normalize_vectors stays False in (b): the literal
    axis-0 normalisation of :190 contradicts the known answers of (a), which pin unit-length rows."""
    import oracle.retrieval_ref as RR
    spec = importlib.util.spec_from_file_location(
        '_ref_retrieval_data', '/root/reference/tests/base_tests/metrics/representation/data.py')
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    known = dict(vectors=d.VECTORS.numpy(), targets=d.TARGETS.numpy(), group_labels=d.GROUP_LABELS.numpy(),
                 queries_idx=d.QUERIES_IDX.numpy(), scores=d.SCORES.numpy(),
                 scores_query_as_relevant=d.SCORES_QUERY_AS_RELEVANT.numpy())
    for table, name in ((d.CLASSIFICATION_ANSWERS, 'classification'), (d.REPRESENTATION_ANSWERS, 'representation'),
                        (d.REPRESENTATION_QUERY_AS_RELEVANT_ANSWERS, 'query_as_relevant')):
        for metric, by_k in table.items():
            known[f'answer__{name}__{metric}'] = np.array([by_k[k] for k in range(1, 7)], dtype=np.float64)
    np.savez_compressed(out_known, **known)
    print('wrote', out_known)

    # ---- stubs of the absent third-party packages --------------------------------------------------------------
    if not hasattr(np, 'bool'):
        np.bool = bool                       # index_base_metric.py:417 uses the pre-1.24 alias

    class _Flat:
        metric = 'IP'

        def __init__(self, dim):
            self.dim, self.x = dim, None

        def add(self, x):
            self.x = np.array(x, dtype=np.float32)

        def search(self, q, k):
            return RR.flat_search(self.x, np.asarray(q), k, self.metric)

    fa = _fake_pkg('faiss')
    fa.IndexFlatIP = type('IndexFlatIP', (_Flat,), dict(metric='IP'))
    fa.IndexFlatL2 = type('IndexFlatL2', (_Flat,), dict(metric='L2'))
    fa.IndexIVFFlat = None
    _fake_pkg('ranx')
    rm = _fake_pkg('ranx.metrics')
    for name, fn in RR.RANX.items():
        setattr(rm, name, (lambda f: lambda qrels, run, k: np.array(
            [f([(int(a), float(b)) for a, b in q], [int(r[0]) for r in rr], k) for q, rr in zip(qrels, run)]))(fn))

    class Metric:
        def __init__(self, **kw):
            pass

        def add_state(self, name, default, dist_reduce_fx=None):
            setattr(self, name, list(default))
    tm = _fake_pkg('torchmetrics')
    tm.Metric = Metric
    _fake_pkg('torchok.metrics', f'{REF}/metrics')
    _load('torchok.metrics.index_base_metric', f'{REF}/metrics/index_base_metric.py')
    rr = _load('torchok.metrics.representation_ranx', f'{REF}/metrics/representation_ranx.py')

    class _Shaped(list):
        @property
        def shape(self):
            return (len(self), 2)
    orig = rr.RanxBasedMeter.process_data_for_metric_func

    def shaped(self, **kw):
        out = orig(self, **kw)
        return [_Shaped(out[0]), out[1], out[2]]
    rr.RanxBasedMeter.process_data_for_metric_func = shaped

    rng = np.random.default_rng(5)
    n, dim = 48, 8
    vectors = rng.standard_normal((n, dim)).astype(np.float32)
    labels = np.repeat(np.arange(6), 8)[rng.permutation(n)]
    groups = rng.integers(0, 3, n)
    # representation data: 5 queries, two of them also relevant to another query
    n_q = 5
    query_idxs = -np.ones(n, dtype=np.int64)
    q_rows = rng.choice(n, n_q, replace=False)
    query_idxs[q_rows] = rng.permutation(n_q)
    scores = np.zeros((n, n_q), dtype=np.float32)
    for c in range(n_q):
        # ragged on purpose: equal-length relevant lists make np.array(..., dtype=object) 2-D and :45 fails on it
        rows = rng.choice(np.setdiff1d(np.arange(n), q_rows[2:]), 4 + 2 * c, replace=False)
        rows = rows[rows != q_rows[list(query_idxs[q_rows]).index(c)]]
        scores[rows, c] = rng.integers(1, 5, len(rows))
    meters = dict(hit_rate=rr.HitAtKMeter, precision=rr.PrecisionAtKMeter, recall=rr.RecallAtKMeter,
                  average_precision=rr.MeanAveragePrecisionAtKMeter, ndcg=rr.NDCGAtKMeter)
    cases, expected = [], []
    for metric, cls in meters.items():
        for ds in ('classification', 'representation'):
            for k, dist, ga, katl in ((1, 'IP', False, False), (3, 'IP', False, False), (5, 'L2', False, False),
                                      (4, 'IP', True, False), (2, 'L2', True, False), (1, 'IP', True, True)):
                m = cls(dataset_type=ds, k=k, metric_distance=dist, group_averaging=ga, k_as_target_len=katl,
                        search_batch_size=7)
                gl = labels if ds == 'classification' else groups
                for lo in range(0, n, 10):
                    sl = slice(lo, lo + 10)
                    if ds == 'classification':
                        m.update(vectors=torch.tensor(vectors[sl]), group_labels=torch.tensor(gl[sl]))
                    else:
                        m.update(vectors=torch.tensor(vectors[sl]), group_labels=torch.tensor(gl[sl]),
                                 query_idxs=torch.tensor(query_idxs[sl]), scores=torch.tensor(scores[sl]))
                ref = m.compute()
                mine = RR.meter_compute(metric, vectors, ds, k=k, group_labels=gl, query_idxs=query_idxs, scores=scores,
                                        metric_distance=dist, group_averaging=ga, k_as_target_len=katl)
                assert abs(ref - mine) < 1e-12, (metric, ds, k, dist, ga, katl, ref, mine)
                cases.append(f'{metric}|{ds}|{k}|{dist}|{int(ga)}|{int(katl)}')
                expected.append(ref)
    np.savez_compressed(out_meters, vectors=vectors, labels=labels, groups=groups, query_idxs=query_idxs, scores=scores,
                        cases=np.array(cases), expected=np.array(expected, dtype=np.float64))
    print(f'wrote {out_meters}: {len(cases)} cases, restatement == reference files: OK')


def golden_pooling(mods, out_path, seed=91):
    """Pooling / PoolingLinear of the reference's OWN poolings/classification/{pooling,linear}.py for every pool type
    ('avg', 'max', 'avgmax', 'catavgmax'; [timm] SelectAdaptivePool2d stubbed by oracle/timm_min.py): outputs, the input
    gradient of a weighted-sum loss and, for PoolingLinear, the weight / bias gradients.  Inputs hold exact ties (bf16
    grid, repeated maxima) so the first-maximum rule of the max gradient is pinned."""
    _, pooling, _, _ = mods
    linear = _load('torchok.models.poolings.classification.linear', f'{REF}/models/poolings/classification/linear.py')
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(5, 16, 7, 7, generator=g) * 2).bfloat16().float()
    x[0, 3, 2, 2] = x[0, 3, 4, 5] = 9.0           # a tie: the earlier pixel takes the whole max gradient
    x[1, :, 0, 0] = 8.0
    x[1, :, 6, 6] = 8.0
    out = {'x': x.numpy()}
    for pt in ('avg', 'max', 'avgmax', 'catavgmax'):
        m = pooling.Pooling(in_channels=16, pooling_type=pt)
        xi = x.clone().requires_grad_(True)
        y = m(xi)
        w = torch.randn(y.shape, generator=g)
        (y * w).sum().backward()
        assert m.out_channels == y.shape[1]
        out.update({f'{pt}_y': y.detach().numpy(), f'{pt}_w': w.numpy(), f'{pt}_dx': xi.grad.numpy()})
        # the restated SelectAdaptivePool2d alone (what oracle/torchok_ref.py uses) gives the same bits
        yo = timm_min.SelectAdaptivePool2d(1, pt, flatten=True)(x)
        assert torch.equal(yo, y.detach())
        pl = linear.PoolingLinear(in_channels=16, out_channels=24, pooling_type=pt)
        with torch.no_grad():
            pl.fc.weight.copy_(torch.randn(pl.fc.weight.shape, generator=g) * 0.2)
            pl.fc.bias.copy_(torch.randn(24, generator=g) * 0.1)
        xi = x.clone().requires_grad_(True)
        z = pl(xi)
        wz = torch.randn(z.shape, generator=g)
        (z * wz).sum().backward()
        out.update({f'{pt}_lin_weight': pl.fc.weight.detach().numpy(), f'{pt}_lin_bias': pl.fc.bias.detach().numpy(),
                    f'{pt}_lin_z': z.detach().numpy(), f'{pt}_lin_wz': wz.numpy(), f'{pt}_lin_dx': xi.grad.numpy(),
                    f'{pt}_lin_dweight': pl.fc.weight.grad.numpy(), f'{pt}_lin_dbias': pl.fc.bias.grad.numpy()})
    np.savez_compressed(out_path, **out)
    print(f'wrote {out_path}: reference Pooling / PoolingLinear, 4 pool types')


def main():
    mods = install_shim()
    gd = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(gd, exist_ok=True)
    # batch/size chosen so that the deepest BatchNorm still sees >= 32 samples per channel (bf16 parity
    # of the HIP path is checked against these same vectors)
    if '--pooling-only' in sys.argv:
        return golden_pooling(mods, os.path.join(gd, 'pooling_modes.npz'))
    if '--retrieval-only' in sys.argv:
        return golden_retrieval(os.path.join(gd, 'retrieval_known_answers.npz'), os.path.join(gd, 'retrieval_meters.npz'))
    if '--metric-only' in sys.argv:
        return golden_metric(os.path.join(gd, 'metric_heads.npz'))
    if '--unsup-only' in sys.argv:
        return golden_unsupervised(os.path.join(gd, 'unsupervised_losses.npz'))
    if '--bce-only' in sys.argv:
        return golden_bce(os.path.join(gd, 'bce_loss.npz'))
    if '--dice-only' in sys.argv:
        return golden_dice(os.path.join(gd, 'dice_loss.npz'))
    if '--ocr-only' in sys.argv:
        return golden_ocr(os.path.join(gd, 'ocr_head_step.npz'))
    if '--davit-only' in sys.argv:
        return golden_davit(os.path.join(gd, 'davit_cls_step.npz'))
    if '--swin-only' in sys.argv:
        return golden_swin(os.path.join(gd, 'swinv2_cls_step.npz'))
    if '--hrnet-only' in sys.argv:
        return golden_hrnet(os.path.join(gd, 'hrnet_seg_step.npz'))
    golden_step(mods, 'resnet18', 10, 8, 96, 11, os.path.join(gd, 'resnet18_cls_step.npz'))
    golden_step(mods, 'resnet50', 16, 8, 128, 12, os.path.join(gd, 'resnet50_cls_step.npz'))
    golden_heads(mods, os.path.join(gd, 'classification_head.npz'))
    golden_pooling(mods, os.path.join(gd, 'pooling_modes.npz'))
    golden_metric(os.path.join(gd, 'metric_heads.npz'))
    golden_hrnet(os.path.join(gd, 'hrnet_seg_step.npz'))
    golden_swin(os.path.join(gd, 'swinv2_cls_step.npz'))
    golden_davit(os.path.join(gd, 'davit_cls_step.npz'))
    golden_ocr(os.path.join(gd, 'ocr_head_step.npz'))
    golden_dice(os.path.join(gd, 'dice_loss.npz'))
    golden_bce(os.path.join(gd, 'bce_loss.npz'))
    golden_unsupervised(os.path.join(gd, 'unsupervised_losses.npz'))
    golden_retrieval(os.path.join(gd, 'retrieval_known_answers.npz'), os.path.join(gd, 'retrieval_meters.npz'))


if __name__ == '__main__':
    main()
