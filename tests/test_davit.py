"""DaViT row: the reference's in-tree Dual Attention Transformer against oracle/davit_ref.py and
tests/golden/davit_cls_step.npz (one training step of the reference's OWN davit.py, tests/golden/gen_golden.py).
Each test runs on the host stand-in and, marked gpu, through libtok_gfx950.so."""
import copy
import os

import numpy as np
import pytest
import torch

import oracle.davit_ref as D
import torchok_amd as T
from helpers import deterministic_state, rel_err

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'davit_cls_step.npz'))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


def _pair(seed=23, **kw):
    m = T.BACKBONES.get('davit_t')(pretrained=False, **kw)
    ref = D.davit_t(**{k: v for k, v in kw.items() if k != 'img_size'})
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    sd = deterministic_state(ref.state_dict(), seed)
    ref.load_state_dict(sd)
    m.load_state_dict(sd)
    return m, ref


def test_structure():
    m, ref = _pair(window_size=4)
    assert m.out_encoder_channels == (96, 192, 384, 768) and m.out_channels == 768
    assert m.get_stages(2) is m                                             # davit.py:527-536
    assert sum(p.numel() for p in D.davit_t().parameters()) == sum(p.numel() for p in m.parameters())
    assert [tuple(getattr(T.BACKBONES.get(n)(pretrained=False), 'embed_dims')) for n in ('davit_s', 'davit_b')] == \
        [(96, 192, 384, 768), (128, 256, 512, 1024)]
    with pytest.raises(RuntimeError):
        T.BACKBONES.get('davit_t')(pretrained=True)


def test_forward_features_and_backward_vs_oracle(dev):
    """Every stage output and every parameter gradient (eval-free: drop_path_rate 0) against the fp32 oracle, with torch's
    own bf16 autocast run of the oracle as the yardstick."""
    m, ref = _pair(window_size=4, drop_path_rate=0.0)
    m.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 3, 128, 128, generator=g)
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
    live = [n for n, v in g32.items() if v is not None]
    assert sorted(n for n, p in m.named_parameters() if p.grad is None) == sorted(n for n in g32 if g32[n] is None)
    assert all('cpe' in n for n in g32 if g32[n] is None)        # ConvPosEnc without activation: inert (davit.py:124-128)
    yard = {n: rel_err(dict(ac.named_parameters())[n].grad, g32[n]) for n in live}
    errs = {n: rel_err(dict(m.named_parameters())[n].grad, g32[n]) for n in live}
    assert np.median(list(errs.values())) < 1.5 * np.median(list(yard.values())) + 1e-2
    bad = [n for n in errs if errs[n] > 1.5 * yard[n] + 0.08]
    assert len(bad) <= 0.05 * len(errs), [(n, errs[n], yard[n]) for n in bad][:8]


def test_conv_pos_enc_with_activation(dev):
    """cpe_act=True: x + GELU(depthwise3x3(x)) before each attention / MLP — forward features and every gradient (the
    cpe parameters now receive one) against the oracle, torch's bf16 autocast run as the yardstick."""
    m, ref = _pair(window_size=4, drop_path_rate=0.0, cpe_act=True)
    m.to(dev).train()
    ref.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 128, 128, generator=g)
    feats = m.forward_features(x.to(dev))
    rfeats = ref.forward_features(x)
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
    assert all(v is not None for v in g32.values()) and all(p.grad is not None for p in m.parameters())
    yard = {n: rel_err(p.grad, g32[n]) for n, p in ac.named_parameters()}
    errs = {n: rel_err(p.grad.float(), g32[n]) for n, p in m.named_parameters()}
    assert np.median(list(errs.values())) < 1.5 * np.median(list(yard.values())) + 1e-2
    bad = [n for n in errs if errs[n] > 1.5 * yard[n] + 0.08]
    assert len(bad) <= 0.05 * len(errs), [(n, errs[n], yard[n]) for n in bad][:8]


def test_reference_shape_test(dev):
    """tests/additional_tests/models/backbones/test_backbone.py (davit_t on 2x3x224x224: window 7 on 56/28/14/7 maps)."""
    m = T.BACKBONES.get('davit_t')(pretrained=False).to(dev).eval()
    x = torch.rand(2, 3, 224, 224).to(dev)
    with torch.no_grad():
        assert tuple(m(x).shape) == (2, 768, 7, 7)
        feats = m.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(2, 3, 224, 224), (2, 96, 56, 56), (2, 192, 28, 28), (2, 384, 14, 14),
                                               (2, 768, 7, 7)]
    with pytest.raises(NotImplementedError, match='multiple of the window'):
        m(torch.rand(1, 3, 160, 160).to(dev))


def test_classification_step_vs_reference_golden(dev, monkeypatch):
    """ClassificationTask(davit_t + Pooling + ClassificationHead) + CE + AdamW, training mode with stochastic depth 0.2:
    one step of the reference's own davit.py (golden) — last feature, logits, loss, every gradient norm, the small
    gradients element-wise.  The per-sample keep/scale factors the reference drew are replayed (a device RNG cannot
    reproduce torch's CPU stream)."""
    from helpers import cls_config
    from torchok_amd.models.backbones import swin as SW
    cfg = cls_config('davit_t', int(GOLD['num_classes']), optimizer='AdamW', opt_params={'lr': 1e-3, 'weight_decay': 0.05},
                     backbone_params=dict(img_size=128, window_size=4, drop_path_rate=0.2), inputs_shape=(3, 128, 128))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    sd = deterministic_state({k: v for k, v in task.state_dict().items() if not k.startswith('input_tensors')},
                             int(GOLD['seed']))
    task.load_state_dict(sd, strict=False)
    task.to(dev)
    x = torch.from_numpy(GOLD['x'].astype(np.float32)).to(dev)
    y = torch.from_numpy(GOLD['y']).to(dev)
    task.eval()
    with torch.no_grad():
        assert rel_err(task.backbone(x).float(), torch.from_numpy(GOLD['eval_last_feature'])) < 2e-2
    task.train()
    queue = [torch.from_numpy(r) for r in GOLD['drop_scales']]

    def replay(self, batch, device):
        return queue.pop(0).to(device)
    monkeypatch.setattr(SW.DropPath, 'sample_scale', replay)
    out = task.training_step({'image': x, 'target': y}, 0)
    assert not queue                                            # as many draws as the reference made, in its order
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
        assert rel_err(params[n].grad.float(), ref) < 0.15, n
    opt.step()
    pn = np.array([float(params[n].detach().double().norm()) for n in names])
    assert np.max(np.abs(pn / GOLD['post_step_norm'] - 1)) < 2e-3
