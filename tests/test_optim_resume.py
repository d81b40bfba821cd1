"""Checkpoint resume and late parameter groups of the fused arena optimizers.

Lightning's `resume_path` (reference `__main__.py:41`) restores the optimizer through `load_state_dict`; the reference's
FreezeUnfreeze callback (`callbacks/freeze_unfreeze.py:51-184`) adds parameter groups after training has started.  Both must
behave exactly like the `torch.optim` classes the reference registers (`optim/optimizers/__init__.py:9-19`)."""
import copy

import pytest
import torch

import torchok_amd as T
from helpers import rel_err


def _plain_params(seed, shapes=((6, 5), (5,), (4, 3, 3, 3), (7,))):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]


def _grads(step, params):
    g = torch.Generator().manual_seed(1000 + step)
    return [torch.randn(p.shape, generator=g) for p in params]


CASES = [('SGD', torch.optim.SGD, dict(lr=0.1, momentum=0.9, weight_decay=1e-4)),
         ('SGD', torch.optim.SGD, dict(lr=0.1, momentum=0.9, nesterov=True)),
         ('Adam', torch.optim.Adam, dict(lr=1e-2)),
         ('AdamW', torch.optim.AdamW, dict(lr=1e-2, weight_decay=0.05)),
         ('RMSprop', torch.optim.RMSprop, dict(lr=1e-2, momentum=0.5, centered=True))]


@pytest.mark.parametrize('name,tcls,kw', CASES)
def test_save_load_continue_equals_torch(fake_backend, name, tcls, kw):
    ours_p, ref_p = _plain_params(0), _plain_params(0)
    ours, ref = T.OPTIMIZERS.get(name)(ours_p, **kw), tcls(ref_p, **kw)

    def run(opt, params, steps):
        for s in steps:
            for p, g in zip(params, _grads(s, params)):
                p.grad = g.clone()
            opt.step()

    run(ours, ours_p, range(2))
    run(ref, ref_p, range(2))
    sd = copy.deepcopy(ours.state_dict())
    assert len(sd['state']) == len(ours_p)
    # a NEW optimizer over NEW parameter objects (what a resumed run has), state restored through load_state_dict
    new_p = [torch.nn.Parameter(p.detach().clone()) for p in ours_p]
    resumed = T.OPTIMIZERS.get(name)(new_p, **kw)
    resumed.load_state_dict(sd)
    run(resumed, new_p, range(2, 4))
    run(ref, ref_p, range(2, 4))
    for a, b in zip(new_p, ref_p):
        assert rel_err(a, b) < 2e-5, (name, kw)
    # the resumed state lives in the arena slots (what the step kernels read), and a second save re-exports live values
    st = resumed.state[new_p[0]]
    key = {'SGD': 'momentum_buffer', 'RMSprop': 'square_avg'}.get(name, 'exp_avg')
    arena = resumed._arenas[0]
    assert st[key].data_ptr() == arena.state_view(0, 0).data_ptr()
    again = resumed.state_dict()['state'][0][key]
    assert rel_err(again, ref.state[ref_p[0]][key]) < 2e-5


def test_load_state_dict_after_first_step_overrides_live_state(fake_backend):
    kw = dict(lr=0.1, momentum=0.9)
    a_p, b_p = _plain_params(1), _plain_params(1)
    a, b = T.OPTIMIZERS.get('SGD')(a_p, **kw), T.OPTIMIZERS.get('SGD')(b_p, **kw)
    for opt, ps in ((a, a_p), (b, b_p)):
        for p, g in zip(ps, _grads(0, ps)):
            p.grad = g.clone()
        opt.step()
    # b takes a's state after having stepped itself (arena already built): its momentum must be a's afterwards
    for p, g in zip(a_p, _grads(7, a_p)):
        p.grad = g.clone()
    a.step()
    b.load_state_dict(copy.deepcopy(a.state_dict()))
    with torch.no_grad():
        for pa, pb in zip(a_p, b_p):
            pb.copy_(pa)
    for opt, ps in ((a, a_p), (b, b_p)):
        for p, g in zip(ps, _grads(8, ps)):
            p.grad = g.clone()
        opt.step()
    for pa, pb in zip(a_p, b_p):
        assert torch.equal(pa, pb)


@pytest.mark.parametrize('name,tcls,kw', CASES[:1] + CASES[2:3])
def test_add_param_group_after_first_step(fake_backend, name, tcls, kw):
    ours_p, ref_p = _plain_params(2), _plain_params(2)
    ours, ref = T.OPTIMIZERS.get(name)(ours_p[:2], **kw), tcls(ref_p[:2], **kw)
    for opt, ps in ((ours, ours_p), (ref, ref_p)):
        for p, g in zip(ps[:2], _grads(0, ps[:2])):
            p.grad = g.clone()
        opt.step()
    gen0 = ours.arena_generation
    ours.add_param_group({'params': ours_p[2:], 'lr': kw['lr'] * 0.5})
    ref.add_param_group({'params': ref_p[2:], 'lr': kw['lr'] * 0.5})
    for s in (1, 2):
        for opt, ps in ((ours, ours_p), (ref, ref_p)):
            for p, g in zip(ps, _grads(s, ps)):
                p.grad = g.clone()
            opt.step()
    for a, b in zip(ours_p, ref_p):
        assert rel_err(a, b) < 2e-5
    assert len(ours._arenas) == 2 and ours.arena_generation == gen0 + 1      # the first group kept its arena and state


def test_adam_capturable_matches_plain_adam(fake_backend):
    """torch.optim.Adam(capturable=True): device-side step count (tok_adam_step_capturable + tok_step_advance) — the same
    trajectory as the host-step form, step for step, and as torch's own AdamW."""
    import torch
    import torchok_amd as T
    torch.manual_seed(0)
    ws = [torch.randn(5, 3), torch.randn(7)]
    gs = [[torch.randn_like(w) for w in ws] for _ in range(4)]

    def run(make):
        ps = [torch.nn.Parameter(w.clone()) for w in ws]
        opt = make(ps)
        for it in range(4):
            for p, g in zip(ps, gs[it]):
                p.grad = g.clone()
            opt.step()
        return [p.detach().clone() for p in ps], opt
    ref, _ = run(lambda ps: torch.optim.AdamW(ps, lr=1e-2, weight_decay=0.05))
    plain, _ = run(lambda ps: T.OPTIMIZERS.get('AdamW')(ps, lr=1e-2, weight_decay=0.05))
    capt, opt = run(lambda ps: T.OPTIMIZERS.get('AdamW')(ps, lr=1e-2, weight_decay=0.05, capturable=True))
    for a, b, c in zip(ref, plain, capt):
        assert torch.allclose(a, b, atol=1e-6) and torch.allclose(b, c, atol=1e-7)
    assert fake_backend.calls.count('adam_step_capturable') == 4
    steps = {int(s['step']) for s in opt.state.values()}
    assert steps == {4}
    # the recorded step survives a state_dict round trip into a plain (host-step) optimizer
    ps2 = [torch.nn.Parameter(w.clone()) for w in ws]
    opt2 = T.OPTIMIZERS.get('AdamW')(ps2, lr=1e-2, weight_decay=0.05)
    opt2.load_state_dict(opt.state_dict())
    for p, g in zip(ps2, gs[0]):
        p.grad = g.clone()
    opt2.step()
    assert {int(s['step']) for s in opt2.state.values()} == {5}
