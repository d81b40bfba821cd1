"""Host logic of the engine on a CPU-only box: tape / gradient fan-in / parameter arena / fused
optimizer bookkeeping / Task API, with the native library replaced by tests/fake_backend.py (the
same C ABI restated on host memory).  Numerics of the real kernels are checked by the -m gpu tests;
here the reference is the fp32 oracle with torch's own bf16-autocast run as the noise yardstick."""
import copy

import pytest
import torch

import oracle.torchok_ref as R
import torchok_amd as T
from helpers import cls_config, copy_state, deterministic_state, rel_err


def _autocast_grads(ref, x, y):
    ref2 = copy.deepcopy(ref)
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o = ref2.forward_with_gt({'image': x, 'target': y})
    torch.nn.functional.cross_entropy(o['prediction'].float(), y).backward()
    return {n: p.grad for n, p in ref2.named_parameters()}


def _pair(backbone='resnet18', classes=10, seed=5, **opt):
    cfg = cls_config(backbone, classes, **opt)
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    ref = R.ClassificationModel(backbone, classes).train()
    ref.load_state_dict(deterministic_state(ref.state_dict(), seed))
    copy_state(ref, task)
    return task, ref


def test_no_backend_no_cpu_path():
    m = T.BACKBONES.get('resnet18')(pretrained=False)
    with pytest.raises(RuntimeError, match='HIP'):
        m(torch.rand(1, 3, 32, 32))


def test_training_step_matches_oracle(fake_backend):
    torch.manual_seed(0)
    task, ref = _pair()
    x, y = torch.randn(8, 3, 64, 64), torch.randint(0, 10, (8,))
    out = task.training_step({'image': x, 'target': y}, 0)
    assert set(out) == {'loss'} and out['loss'].dim() == 0
    fwd = task.forward_with_gt({'image': x, 'target': y})
    assert set(fwd) == {'embeddings', 'prediction', 'target'}
    assert fwd['embeddings'].shape == (8, 512) and fwd['prediction'].shape == (8, 10)
    out['loss'].backward()
    ref_loss, _ = R.training_step(ref, {'image': x, 'target': y}, None)
    assert abs(float(out['loss']) - float(ref_loss)) < 2e-2 * max(1, abs(float(ref_loss)))
    ac = _autocast_grads(ref, x, y)
    rp = dict(ref.named_parameters())
    for n, p in task.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
        assert rel_err(p.grad, rp[n].grad) < 1.5 * rel_err(ac[n], rp[n].grad) + 1e-2, n
    # residual fan-in went through the kernels' accumulate path / in-place donation, not torch adds
    assert fake_backend.calls.count('conv_dgrad') == 19 + 1      # 20 convs (stem needs no data gradient) + fc
    assert fake_backend.calls.count('conv_wgrad') == 20 + 1      # + fc


def test_fused_sgd_equals_torch_sgd(fake_backend):
    torch.manual_seed(1)
    task, ref = _pair()
    opt = task.configure_optimizers()[0]['optimizer']
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    rp = dict(ref.named_parameters())
    for it in range(3):
        out = task.training_step({'image': x, 'target': y}, it)
        opt.zero_grad()
        out['loss'].backward()
        for n, p in task.named_parameters():        # same gradients on both sides
            rp[n].grad = p.grad.detach().clone().contiguous()
        opt.step()
        ropt.step()
        for n, p in task.named_parameters():
            assert rel_err(p, rp[n]) < 1e-6, (it, n)
    # one flat launch per step for the single param group (reference constructor.py:151-152)
    assert fake_backend.calls.count('sgd_step') == 3
    arena = opt._arenas[0]
    assert all(arena.owns_data(i) for i in range(len(arena.params)))
    assert all(p.grad.data_ptr() == arena.grad_view(i).data_ptr() for i, p in enumerate(arena.params))
    sd = opt.state_dict()
    assert len(sd['state']) == len(arena.params) and 'momentum_buffer' in sd['state'][0]


def test_fused_adam_equals_torch_adam(fake_backend):
    torch.manual_seed(2)
    task, ref = _pair(optimizer='Adam', opt_params={'lr': 1e-3})
    opt = task.configure_optimizers()[0]['optimizer']
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    rp = dict(ref.named_parameters())
    for it in range(2):
        out = task.training_step({'image': x, 'target': y}, it)
        opt.zero_grad()
        out['loss'].backward()
        for n, p in task.named_parameters():
            rp[n].grad = p.grad.detach().clone().contiguous()
        opt.step()
        ropt.step()
        for n, p in task.named_parameters():
            assert rel_err(p, rp[n]) < 1e-5, (it, n)


@pytest.mark.parametrize('kw', [dict(lr=1e-2), dict(lr=1e-2, momentum=0.9, weight_decay=1e-4), dict(lr=5e-3, centered=True, alpha=0.9),
                                dict(lr=5e-3, centered=True, momentum=0.5, eps=1e-6)])
def test_fused_rmsprop_equals_torch_rmsprop(fake_backend, kw):
    _rmsprop_case('cpu', kw)


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(lr=1e-2), dict(lr=5e-3, centered=True, momentum=0.5, weight_decay=1e-4)])
def test_fused_rmsprop_equals_torch_rmsprop_gpu(kw):
    _rmsprop_case('cuda', kw)


def _rmsprop_case(dev, kw):
    torch.manual_seed(3)
    task, ref = _pair(optimizer='RMSprop', opt_params=kw)
    task.to(dev)
    opt = task.configure_optimizers()[0]['optimizer']
    assert type(opt).__name__ == 'RMSprop' and type(opt).__module__.startswith('torchok_amd')
    ropt = torch.optim.RMSprop(ref.parameters(), **kw)
    x, y = torch.randn(4, 3, 32, 32).to(dev), torch.randint(0, 10, (4,)).to(dev)
    rp = dict(ref.named_parameters())
    for it in range(3):
        out = task.training_step({'image': x, 'target': y}, it)
        opt.zero_grad()
        out['loss'].backward()
        for n, p in task.named_parameters():        # same gradients on both sides
            rp[n].grad = p.grad.detach().float().cpu().clone().contiguous()
        opt.step()
        ropt.step()
        for n, p in task.named_parameters():
            assert rel_err(p, rp[n]) < 2e-5, (it, n)
    sd = opt.state_dict()
    assert 'square_avg' in sd['state'][0] and ('momentum_buffer' in sd['state'][0]) == (kw.get('momentum', 0) > 0)
    assert ('grad_avg' in sd['state'][0]) == bool(kw.get('centered', False))


@pytest.mark.parametrize('name,kw', [('SGD', dict(lr=0.05, momentum=0.9, weight_decay=1e-4)), ('AdamW', dict(lr=1e-3, weight_decay=0.05))])
def test_cached_runs_follow_a_changing_gradient_pattern(fake_backend, name, kw):
    """The optimizers answer "which runs of parameters have a gradient, with which state key" from the previous step while the
    pattern of gradients repeats (no per-parameter walk: optim/optimizers.py _runs).  Seven steps against torch.optim fed the
    same gradients: the whole arena (cached from step 2 on), then one parameter without a gradient for two steps (runs are
    re-cut, that parameter's Adam step count stays behind), a foreign gradient tensor (adopted into the arena), and back."""
    torch.manual_seed(5)
    task, ref = _pair(optimizer=name, opt_params=kw)
    opt = task.configure_optimizers()[0]['optimizer']
    ropt = getattr(torch.optim, name)(ref.parameters(), **kw)
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    rp = dict(ref.named_parameters())
    names = [n for n, _ in task.named_parameters()]
    skip, foreign = names[len(names) // 2], names[3]
    walks = []
    from torchok_amd.engine.arena import ParamArena
    real_adopt = ParamArena.adopt_grad

    def counting(self_, i):
        walks.append(i)
        return real_adopt(self_, i)
    ParamArena.adopt_grad = counting
    try:
        per_step = []
        for it in range(7):
            out = task.training_step({'image': x, 'target': y}, it)
            opt.zero_grad()
            out['loss'].backward()
            tp = dict(task.named_parameters())
            if it in (3, 4):
                tp[skip].grad = None
            if it == 5:
                tp[foreign].grad = tp[foreign].grad.detach().clone()          # a gradient tensor outside the arena
            for n, p in tp.items():
                rp[n].grad = None if p.grad is None else p.grad.detach().float().clone().contiguous()
            walks.clear()
            opt.step()
            per_step.append(len(walks))
            ropt.step()
            for n, p in task.named_parameters():
                assert rel_err(p, rp[n]) < 2e-5, (it, n)
    finally:
        ParamArena.adopt_grad = real_adopt
    n = len(names)
    # walk on the first step, on every change of the pattern (steps 3, 5, 6) and never in between (1, 2, 4)
    assert per_step[0] == n and per_step[1] == 0 and per_step[2] == 0, per_step
    assert per_step[3] == n and per_step[4] == 0 and per_step[5] == n and per_step[6] == n, per_step
    if name == 'AdamW':
        st = opt.state_dict()['state']
        steps = sorted({int(v['step']) for v in st.values()})
        assert steps == [5, 7], steps          # the skipped parameter missed two steps


def test_frozen_parameters_get_no_grad_and_are_skipped(fake_backend):
    """FreezeUnfreeze-style freezing (reference callbacks/freeze_unfreeze.py): frozen params keep
    grad None, stay in the optimizer, and are not updated (torch semantics: grad None => skip)."""
    torch.manual_seed(3)
    task, _ = _pair()
    for p in task.backbone.get_stages(1).parameters():
        p.requires_grad_(False)
    frozen = {n: p.detach().clone() for n, p in task.named_parameters() if not p.requires_grad}
    assert len(frozen) > 0
    opt = task.configure_optimizers()[0]['optimizer']
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    out = task.training_step({'image': x, 'target': y}, 0)
    out['loss'].backward()
    opt.step()
    for n, p in task.named_parameters():
        if n in frozen:
            assert p.grad is None and torch.equal(p.detach(), frozen[n]), n
        else:
            assert p.grad is not None, n
    assert fake_backend.calls.count('sgd_step') == 1   # trainable params are one contiguous run


def test_eval_and_no_grad_paths(fake_backend):
    torch.manual_seed(4)
    task, ref = _pair()
    task.eval(), ref.eval()
    x = torch.randn(2, 3, 64, 64)
    with torch.no_grad():
        got = task(x)
        want = ref.head.fc(ref.pooling(ref.backbone(x)))
    assert got.shape == (2, 10) and rel_err(got.float(), want) < 5e-2
    assert int(task.backbone.bn1.num_batches_tracked) == 0       # eval: running stats untouched
    feats = task.backbone.forward_features(x)
    assert [tuple(f.shape) for f in feats] == [(2, 3, 64, 64), (2, 64, 32, 32), (2, 64, 16, 16), (2, 128, 8, 8),
                                               (2, 256, 4, 4), (2, 512, 2, 2)]


def test_multi_output_region_backward(fake_backend):
    """forward_features returns 5 feature maps from ONE region; gradients arriving at inner features
    are merged with the in-region consumers' contributions."""
    torch.manual_seed(5)
    m = T.BACKBONES.get('resnet18')(pretrained=False, zero_init_last=False).train()
    x = torch.randn(2, 3, 64, 64)
    feats = m.forward_features(x)
    loss = sum(f.float().mean() for f in feats[1:])
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_state_dict_roundtrip_after_arena(fake_backend, tmp_path):
    torch.manual_seed(6)
    task, _ = _pair()
    opt = task.configure_optimizers()[0]['optimizer']
    x, y = torch.randn(2, 3, 32, 32), torch.randint(0, 10, (2,))
    task.training_step({'image': x, 'target': y}, 0)['loss'].backward()
    opt.step()
    torch.save(task.state_dict(), tmp_path / 'ck.pt')
    sd = torch.load(tmp_path / 'ck.pt')
    task2, _ = _pair(seed=99)
    task2.load_state_dict(sd)
    for (n, a), (_, b) in zip(task.state_dict().items(), task2.state_dict().items()):
        assert torch.equal(a, b), n
    assert sd['backbone.conv1.weight'].shape == (64, 3, 7, 7)


def test_packs_follow_the_masters(fake_backend):
    """bf16 MFMA operand packs vs fp32 masters: the fused optimizer refreshes all of them in ONE launch after
    its step; a forward re-packs a weight individually only if torch saw it change (`_version`) or after
    invalidate_packs()."""
    from torchok_amd.engine import functional as EF
    torch.manual_seed(3)
    task, _ = _pair()
    opt = task.configure_optimizers()[0]['optimizer']
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    conv = task.backbone.layer2[0].conv1

    def fwd_pack_in_sync():
        pk = EF._packs[id(conv.weight)][1]
        want = conv.weight.detach().permute(0, 2, 3, 1).to(torch.bfloat16)
        return torch.equal(pk.fwd[:, :, :want.shape[2], :], want) and pk.synced == conv.weight._version

    for it in range(3):
        out = task.training_step({'image': x, 'target': y}, it)
        opt.zero_grad()
        out['loss'].backward()
        n_single = fake_backend.calls.count('pack_weights_batched')
        opt.step()
        assert fake_backend.calls.count('pack_weights_batched') == n_single + (1 if it > 0 else 0) or it == 0
        if it > 0:     # step 0 re-homes the masters into the arena: packs are re-derived by the next forward
            assert fwd_pack_in_sync()
    assert fake_backend.calls.count('pack_weights_batched') >= 2
    # a torch-visible write is caught by the next forward ...
    with torch.no_grad():
        conv.weight.mul_(1.5)
    assert not fwd_pack_in_sync()
    task.training_step({'image': x, 'target': y}, 9)
    assert fwd_pack_in_sync()
    # ... a write behind torch's back needs invalidate_packs()
    conv.weight.data.mul_(2.0)
    EF.invalidate_packs()
    task.training_step({'image': x, 'target': y}, 10)
    assert fwd_pack_in_sync()


def test_fused_bn_finalize_path(fake_backend, monkeypatch):
    """Opt-in engine path over tok_conv_fwd_bn / tok_conv_dgrad_bn (off by default: measured slower)."""
    from torchok_amd.engine import functional as EF
    monkeypatch.setattr(EF, 'FUSE_BN_FINALIZE', True)
    torch.manual_seed(0)
    task, ref = _pair()
    x, y = torch.randn(8, 3, 64, 64), torch.randint(0, 10, (8,))
    out = task.training_step({'image': x, 'target': y}, 0)
    out['loss'].backward()
    ref_loss, _ = R.training_step(ref, {'image': x, 'target': y}, None)
    assert abs(float(out['loss']) - float(ref_loss)) < 2e-2 * max(1, abs(float(ref_loss)))
    ac = _autocast_grads(ref, x, y)
    rp = dict(ref.named_parameters())
    for n, p in task.named_parameters():
        assert rel_err(p.grad, rp[n].grad) < 1.5 * rel_err(ac[n], rp[n].grad) + 1e-2, n
    assert task.backbone.bn1.num_batches_tracked.item() == 1


def test_strided_projection_runs_pointwise_and_parks_its_gradient(fake_backend, monkeypatch):
    """[timm] downsample_conv with stride 2 (resnet.py): conv1x1/s2(x) == conv1x1(x[:, ::2, ::2]).  The half-resolution
    gradient is absorbed by conv1's data gradient (tok_conv_dgrad_subacc) where that layer qualifies and expanded by
    tok_subsample2_bwd where it does not; gradients agree with the oracle either way."""
    import torchok_amd.engine.functional as EF
    monkeypatch.setattr(EF, 'SUBSAMPLE_MIN_ROWS', 0)
    torch.manual_seed(3)
    task, ref = _pair('resnet50', 10, seed=9)
    x, y = torch.randn(4, 3, 64, 64), torch.randint(0, 10, (4,))
    out = task.training_step({'image': x, 'target': y}, 0)
    out['loss'].backward()
    ref_loss, _ = R.training_step(ref, {'image': x, 'target': y}, None)
    assert abs(float(out['loss']) - float(ref_loss)) < 2e-2 * max(1, abs(float(ref_loss)))
    ac = _autocast_grads(ref, x, y)
    rp = dict(ref.named_parameters())
    for n, p in task.named_parameters():
        assert p.grad is not None, n
        assert rel_err(p.grad, rp[n].grad) < 1.5 * rel_err(ac[n], rp[n].grad) + 1e-2, n
    calls = fake_backend.calls
    assert calls.count('subsample2_fwd') == 3                       # layers 2.0, 3.0, 4.0
    # the fake backend's closers are the layers with c % 64 == 0: all three conv1's absorb the parked gradient
    assert calls.count('dgrad_subacc') == 3 and calls.count('subsample2_bwd') == 0

    # a consumer that cannot absorb it: the stand-alone scatter takes over (same gradients)
    monkeypatch.setattr(type(fake_backend), 'tok_conv_dgrad_subacc_ok', lambda self, d: 0)
    task2, _ = _pair('resnet50', 10, seed=9)
    fake_backend.calls.clear()
    out2 = task2.training_step({'image': x, 'target': y}, 0)
    out2['loss'].backward()
    assert fake_backend.calls.count('subsample2_bwd') == 3 and fake_backend.calls.count('dgrad_subacc') == 0
    g1 = dict(task.named_parameters())
    for n, p in task2.named_parameters():
        assert rel_err(p.grad, g1[n].grad) < 1e-6, n
