"""SwinV2 row: SwinTransformerV2 backbone (+ ClassificationTask) against oracle/swin_ref.py and
tests/golden/swinv2_cls_step.npz (one training step of the reference's own swin.py, tests/golden/gen_golden.py).
Each test runs on the host stand-in and, marked gpu, through libtok_gfx950.so."""
import copy
import os

import numpy as np
import pytest
import torch

import oracle.swin_ref as S
import torchok_amd as T
from helpers import deterministic_state, rel_err

KW = dict(img_size=64, window_size=4, depths=(2, 2, 2, 2), drop_path_rate=0.0)


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _pair(seed=17, **kw):
    kw = dict(KW, **kw)
    m = T.BACKBONES.get('swinv2_custom')(**kw)
    ref = S.SwinV2(**kw)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    sd = deterministic_state(ref.state_dict(), seed)
    ref.load_state_dict(sd)
    m.load_state_dict(sd)
    return m, ref


def test_structure_and_index_buffers():
    m, ref = _pair()
    for (n1, b1), (n2, b2) in zip(m.named_buffers(), ref.named_buffers()):
        assert n1 == n2 and torch.equal(b1, b2), n1          # attn_mask, relative_position_index, coords table: exact
    assert sorted(m.no_weight_decay()) == sorted(
        ['absolute_pos_embed'] + [n for n, _ in ref.named_modules() if 'cpb_mlp' in n or 'logit_scale' in n])
    assert m.out_encoder_channels == (96, 192, 384, 768) and m.out_channels == 768
    assert len(m.get_stages(2)) == 4
    big = S.SwinV2(img_size=224, window_size=7)
    assert sum(p.numel() for p in big.parameters()) == 27579498            # 28.35 M with the 1000-way head


def test_forward_features_and_backward_vs_oracle(dev):
    m, ref = _pair()
    m.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 3, 64, 64, generator=g)
    feats = m.forward_features(x.to(dev))
    rfeats = ref.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [tuple(f.shape) for f in rfeats]
    for i, (a, b) in enumerate(zip(feats[1:], rfeats[1:])):
        assert rel_err(a.float(), b) < 2e-2, i
    w = [torch.randn(f.shape, generator=g) for f in rfeats[1:]]
    sum((f.float() * wi.to(dev)).sum() for f, wi in zip(feats[1:], w)).backward()
    sum((f * wi).sum() for f, wi in zip(rfeats[1:], w)).backward()
    ac = copy.deepcopy(ref)
    ac.zero_grad()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        af = ac.forward_features(x)
    sum((f.float() * wi).sum() for f, wi in zip(af[1:], w)).backward()
    g32 = {n: p.grad for n, p in ref.named_parameters()}
    yard = {n: rel_err(p.grad, g32[n]) for n, p in ac.named_parameters()}
    errs = {n: rel_err(p.grad, g32[n]) for n, p in m.named_parameters()}
    assert all(p.grad is not None and p.grad.shape == p.shape for p in m.parameters())
    assert np.median(list(errs.values())) < 1.5 * np.median(list(yard.values())) + 1e-2
    bad = [n for n in errs if errs[n] > 1.5 * yard[n] + 0.08]
    assert len(bad) <= 0.05 * len(errs), [(n, errs[n], yard[n]) for n in bad][:8]


def test_reference_shape_tests(dev):
    """tests/additional_tests/models/backbones/test_backbone.py:161-178 (swinv2_tiny_window16_256 on 2x3x256x256)."""
    m = T.BACKBONES.get('swinv2_tiny_window16_256')(pretrained=False).to(dev).eval()
    x = torch.rand(2, 3, 256, 256).to(dev)
    with torch.no_grad():
        assert tuple(m(x).shape) == (2, 768, 8, 8)
        feats = m.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(2, 3, 256, 256), (2, 96, 64, 64), (2, 192, 32, 32), (2, 384, 16, 16),
                                               (2, 768, 8, 8)]


def test_stochastic_depth_scales_whole_samples(dev):
    m, ref = _pair(drop_path_rate=0.5)
    m.to(dev).train()
    x = torch.randn(8, 3, 64, 64)
    torch.manual_seed(0)
    a = m(x.to(dev)).float().cpu()
    torch.manual_seed(1)
    b = m(x.to(dev)).float().cpu()
    assert not torch.allclose(a, b)                 # different drop masks
    m.eval()
    with torch.no_grad():
        c, d = m(x.to(dev)).float().cpu(), ref.eval()(x)
    assert rel_err(c, d) < 2e-2                     # eval: no drop


GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'swinv2_cls_step.npz'))


def test_classification_step_vs_reference_golden(dev):
    """ClassificationTask(swinv2_custom + Pooling + ClassificationHead) + CE + AdamW: one step of the reference's own
    swin.py (golden) — logits, loss, every gradient norm, parameters after the optimizer step."""
    from helpers import cls_config
    cfg = cls_config('swinv2_custom', int(GOLD['num_classes']), optimizer='AdamW',
                     opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                     backbone_params=dict(img_size=64, window_size=4, depths=[2, 2, 2, 2], drop_path_rate=0.0),
                     inputs_shape=(3, 64, 64))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')},
                             int(GOLD['seed']))
    task.load_state_dict(sd, strict=False)
    task.to(dev).train()
    x = torch.from_numpy(GOLD['x'].astype(np.float32)).to(dev)
    y = torch.from_numpy(GOLD['y']).to(dev)
    last = task.backbone(x)
    assert rel_err(last.float(), torch.from_numpy(GOLD['last_feature'])) < 2e-2
    out = task.training_step({'image': x, 'target': y}, 0)
    pred = task.forward_with_gt({'image': x, 'target': y})['prediction']
    assert rel_err(pred.float(), torch.from_numpy(GOLD['prediction'])) < 3e-2
    assert abs(float(out['loss'].detach()) - float(GOLD['loss'])) < 2e-2 * float(GOLD['loss'])
    opt = task.configure_optimizers()[0]['optimizer']
    opt.zero_grad()
    out['loss'].backward()
    names = [str(n) for n in GOLD['param_names']]
    params = dict(task.named_parameters())
    assert sorted(n for n, p in params.items() if p.grad is None) == sorted(str(n) for n in GOLD['no_grad_names'])
    gn = np.array([float(params[n].grad.detach().double().norm()) for n in names])
    assert np.median(np.abs(gn / GOLD['grad_norm'] - 1)) < 0.05
    for n in (str(s) for s in GOLD['small_names']):
        ref = torch.from_numpy(GOLD['grad__' + n])
        assert rel_err(params[n].grad.float(), ref) < 0.15 + 0.02 / (float(ref.norm()) + 1e-9) * 1e-3, n
    opt.step()
    pn = np.array([float(params[n].detach().double().norm()) for n in names])
    assert np.abs(pn / GOLD['post_step_norm'] - 1).max() < 2e-3


@pytest.mark.gpu
def test_training_loop_learns_and_side_stream_is_transparent(monkeypatch):
    """30 AdamW steps on a fixed mini-batch: the loss falls, and the parameters are bit-identical with the parameter-gradient
    work on the side stream or on the main one (deterministic kernels: a cross-stream race would show)."""
    from helpers import cls_config
    from torchok_amd.engine import functional as EF
    finals, losses = [], []
    for side in (True, False):
        monkeypatch.setattr(EF, 'WGRAD_SIDE_STREAM', side)
        cfg = cls_config('swinv2_custom', 4, optimizer='AdamW', opt_params={'lr': 2e-3, 'weight_decay': 0.01},
                         backbone_params=dict(img_size=64, window_size=4, depths=[2, 2, 2, 2], drop_path_rate=0.0),
                         inputs_shape=(3, 64, 64))
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 23)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator().manual_seed(3)
        x = torch.randn(16, 3, 64, 64, generator=g).cuda()
        y = torch.randint(0, 4, (16,), generator=g).cuda()
        hist = []
        for it in range(30):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad(set_to_none=True)
            out['loss'].backward()
            opt.step()
            hist.append(out['loss'].detach())
        torch.cuda.synchronize()
        losses.append([float(v) for v in hist])
        finals.append({n: p.detach().clone() for n, p in task.named_parameters()})
    assert losses[0][-1] < 0.8 * losses[0][0], (losses[0][0], losses[0][-1])
    assert losses[0] == losses[1]
    for n in finals[0]:
        assert torch.equal(finals[0][n], finals[1][n]), n


@pytest.mark.gpu
def test_fused_activation_epilogues_leave_training_bit_identical(monkeypatch):
    """fc1 + GELU in one launch and GELU' in fc2's dgrad epilogue (engine.transformer.FUSE_ACT) against the separate
    activation launches: 5 AdamW steps, identical losses and parameters."""
    from helpers import cls_config
    from torchok_amd.engine import transformer as ET
    finals = []
    for fuse in (True, False):
        monkeypatch.setattr(ET, 'FUSE_ACT', fuse)
        cfg = cls_config('swinv2_custom', 5, optimizer='AdamW', opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                         backbone_params=dict(img_size=64, window_size=4, depths=[2, 2, 2, 2], drop_path_rate=0.0),
                         inputs_shape=(3, 64, 64))
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 9)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 5, (8,), generator=g).cuda()
        losses = []
        for it in range(5):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad()
            out['loss'].backward()
            opt.step()
            losses.append(float(out['loss'].detach()))
        finals.append((losses, {n: p.detach().clone() for n, p in task.named_parameters()}))
    assert finals[0][0] == finals[1][0]
    for n in finals[0][1]:
        assert torch.equal(finals[0][1][n], finals[1][1][n]), n


@pytest.mark.gpu
def test_fused_mlp_leaves_training_bit_identical(monkeypatch):
    """The whole Mlp from one launch (engine.transformer.FUSE_MLP, csrc/mlp_fused.hip) against fc1 + GELU and fc2 as two
    launches: 5 AdamW steps on widths 96 / 192 / 384 (served) and 768 (not served): identical losses and parameters (saved
    pre / act rows, unfused weight gradients on both sides)."""
    from helpers import cls_config, rel_err
    from torchok_amd.engine import transformer as ET
    from torchok_amd import _C
    finals = []
    for fuse in (True, False):
        monkeypatch.setattr(ET, 'FUSE_MLP', fuse)
        cfg = cls_config('swinv2_custom', 5, optimizer='AdamW', opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                         backbone_params=dict(img_size=64, window_size=4, depths=[2, 2, 2, 2], drop_path_rate=0.0),
                         inputs_shape=(3, 64, 64))
        task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
        sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')}, 9)
        task.load_state_dict(sd, strict=False)
        task.cuda().train()
        opt = task.configure_optimizers()[0]['optimizer']
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 5, (8,), generator=g).cuda()
        losses, grads0 = [], None
        for it in range(5):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad()
            out['loss'].backward()
            if it == 0:
                grads0 = {n: p.grad.detach().clone() for n, p in task.named_parameters() if p.grad is not None}
            opt.step()
            losses.append(float(out['loss'].detach()))
        finals.append((losses, {n: p.detach().clone() for n, p in task.named_parameters()}, grads0))
    assert finals[0][0] == finals[1][0]
    for n in finals[0][1]:
        assert torch.equal(finals[0][1][n], finals[1][1][n]), n



def test_fused_mlp_records_the_tape_of_the_separate_launches(monkeypatch, fake_backend):
    """Host logic (CPU stand-in for the library): mlp_module's served branch launches once and leaves the same outputs,
    input gradient and parameter gradients as fc1 + GELU / fc2; an evaluation pass asks for no saved rows."""
    from torchok_amd.engine import transformer as ET
    from torchok_amd.engine.core import Region
    from torchok_amd.models.backbones.swin import Mlp
    fake = fake_backend
    if True:
        res = []
        for fuse in (True, False):
            monkeypatch.setattr(ET, 'FUSE_MLP', fuse)
            torch.manual_seed(3)
            m = Mlp(96, 384)
            xt = torch.randn(40, 96).to(torch.bfloat16).requires_grad_(True)
            gd = torch.randn(40, 96).to(torch.bfloat16)
            fake.calls.clear()
            r = Region()
            y = r.output(m.run(r, r.input(xt)))
            assert ('mlp_fwd' in fake.calls) == fuse
            y.backward(gd)
            assert ('mlp_bwd_dx' in fake.calls) == fuse
            res.append((y.detach().clone(), xt.grad.clone(), [p.grad.clone() for p in m.parameters()]))
            with torch.no_grad():
                r2 = Region()
                y2 = r2.output(m.run(r2, r2.input(xt.detach())))
            assert torch.equal(y2, y.detach())
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        for a, b in zip(res[0][2], res[1][2]):
            assert torch.equal(a, b)
