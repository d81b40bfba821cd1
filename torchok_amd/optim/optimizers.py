"""Fused flat-arena optimizers registered under the reference's names
(``torchok/optim/optimizers/__init__.py:11,13,18``: SGD, Adam, AdamW) with the constructor
signatures of torch.optim, so ``Constructor.create_optimizer`` (constructor.py:151-158) drives them
unchanged: ``OPTIMIZERS.get(name)(parameters, **optimizer_cfg)``.

One tok_sgd_step / tok_adam_step launch updates a whole run of consecutive parameters in the
arena (engine/arena.py).  Semantics follow torch: parameters whose ``.grad`` is None are skipped,
momentum buffers start as a copy of the first gradient, Adam keeps a per-parameter step count.
"""
from typing import List

import torch
from torch.optim import Optimizer

from .. import _C
from ..constructor import OPTIMIZERS
from ..engine.arena import ParamArena
from ..engine.core import ptr, require_device, stream_ptr


class _ArenaOptimizer(Optimizer):
    _n_state = 0

    def _new_arena(self, group):
        params = [p for p in group['params']]
        if not params:
            return None
        for p in params:
            require_device(p)
        return ParamArena(params, self._n_state)

    def _build(self):
        """(Re)home every group in an arena; optimizer state that already exists (a resumed checkpoint, a rebuild after
        `module.to()`) is copied into the arena's state slots and `self.state` is re-pointed at them."""
        old_state = {id(p): dict(s) for p, s in self.state.items()}
        self._arenas: List[ParamArena] = [self._new_arena(g) for g in self.param_groups]
        self._built = True
        self.arena_generation = getattr(self, 'arena_generation', 0) + 1   # dist/ddp.py re-cuts its buckets on a change
        self._restore_state(old_state)

    def _ensure_built(self):
        if not getattr(self, '_built', False):
            self._build()
            return
        # a module.to()/load that replaced .data detaches parameters from the arena: rebuild
        for g, a in zip(self.param_groups, self._arenas):
            if a is not None and (len(a.params) != len(g['params']) or not a.owns_all()):
                self._build()
                return
        if len(self._arenas) < len(self.param_groups):
            # add_param_group() after the first step (the reference's FreezeUnfreeze callback does it): the new groups get
            # arenas of their own, the existing ones keep theirs (and their state)
            for g in self.param_groups[len(self._arenas):]:
                self._arenas.append(self._new_arena(g))
            self.arena_generation += 1

    def _restore_state(self, old_state):
        pass

    def load_state_dict(self, state_dict):
        """torch's loader leaves stand-alone copies of the state tensors in `self.state`; the step kernels read the
        arena slots, so the loaded moments are copied there and `self.state` is re-pointed at the slots."""
        super().load_state_dict(state_dict)
        self._built = False      # re-homed (and the loaded moments copied into the arena slots) by the next step() / begin_step():
        #                          loading before .cuda() is legal, as with torch.optim
        if all(p.is_cuda for g in self.param_groups for p in g['params']):
            self._ensure_built()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._built = False     # an unpickled optimizer re-homes its parameters at the next step

    def _repack(self):
        """The step kernels moved the fp32 masters: refresh every bf16 MFMA operand pack in one launch."""
        from ..engine.functional import repack_after_step
        if not hasattr(self, '_pack_cache'):
            self._pack_cache = {}
        if self._pack_cache.get('gen') != self.arena_generation:       # (the weight list: 1.1 ms per HRNet-W48 step when rebuilt every time)
            self._pack_cache['gen'] = self.arena_generation
            self._pack_cache['params'] = [p for a in self._arenas if a is not None for p in a.params if p.dim() >= 2]
            self._pack_cache['sig'] = tuple(a.master.data_ptr() for a in self._arenas if a is not None)
            self._pack_cache['tbl'] = {}
        repack_after_step(self._pack_cache['params'], self._pack_cache['tbl'], self._pack_cache['sig'])

    def _runs(self, arena: ParamArena, key_fn):
        """Maximal runs of consecutive parameters that have a gradient and share key_fn(i).  The steady state of a training
        loop — the same parameters have gradients as in the previous step, every one of them in its arena slot — is
        recognised from one pass of identity checks and answered from the runs the previous step left behind
        (`_remember_runs`: the keys as they are AFTER that step); the per-parameter walk (adopt_grad, optimizer-state lookups:
        2 ms per HRNet-W48 step) runs only when the pattern changes."""
        flags = arena.grad_flags()
        cached = getattr(arena, '_runs_cache', None)
        if cached is not None and 2 not in flags and cached[0] == flags:
            return list(cached[1])
        arena._runs_cache = None
        arena._runs_flags = flags if 2 not in flags else None
        runs = []
        cur = None
        for i in range(len(arena.params)):
            if not arena.adopt_grad(i):
                cur = None
                continue
            k = key_fn(i)
            if cur is not None and cur[2] == k:
                cur[1] = i
            else:
                cur = [i, i, k]
                runs.append(cur)
        return [(a, b, k) for a, b, k in runs]

    @staticmethod
    def _remember_runs(arena: ParamArena, runs_after):
        """runs_after: the runs of the step that just ran with the keys they have NOW.  Valid for the next step iff the
        gradient pattern repeats (checked by _runs)."""
        flags = getattr(arena, '_runs_flags', None)
        cached = getattr(arena, '_runs_cache', None)
        if cached is not None:
            arena._runs_cache = (cached[0], runs_after)
        elif flags is not None:
            arena._runs_cache = (flags, runs_after)
            arena._runs_flags = None

    def zero_grad(self, set_to_none: bool = True):
        # keeps torch semantics (default: grads become None; the arena slots are simply rewritten
        # by the next backward — no memset pass over the gradient arena)
        if not set_to_none or not getattr(self, '_built', False):
            return super().zero_grad(set_to_none=set_to_none)
        for g in self.param_groups:         # torch's loop does the same per parameter behind several checks (0.6 ms per HRNet step)
            for p in g['params']:
                p.grad = None


@OPTIMIZERS.register_class
class SGD(_ArenaOptimizer):
    _n_state = 1

    def __init__(self, params, lr=1e-3, momentum=0., dampening=0., weight_decay=0., nesterov=False, *,
                 maximize: bool = False, foreach=None, differentiable: bool = False, fused=None):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if momentum < 0.0:
            raise ValueError(f'Invalid momentum value: {momentum}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError('Nesterov momentum requires a momentum and zero dampening')
        if differentiable:
            raise NotImplementedError('differentiable optimizers are not supported')
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                        nesterov=nesterov, maximize=maximize)
        super().__init__(params, defaults)

    def _restore_state(self, old_state):
        for a in self._arenas:
            if a is None:
                continue
            for i, p in enumerate(a.params):
                s = old_state.get(id(p))
                if s and 'momentum_buffer' in s and s['momentum_buffer'] is not None:
                    a.state_view(0, i).copy_(s['momentum_buffer'])
                    self.state[p]['momentum_buffer'] = a.state_view(0, i)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_built()
        lib, st = _C.lib(), stream_ptr()
        for group, arena in zip(self.param_groups, self._arenas):
            if arena is None:
                continue
            mom = float(group['momentum'])

            def has_buf(i, arena=arena):
                return self.state[arena.params[i]].get('momentum_buffer') is not None
            runs = self._runs(arena, has_buf)
            for a, b, inited in runs:
                off = arena.offsets[a]
                count = arena.padded_end(b) - off
                _C.check(lib.tok_sgd_step(ptr(arena.master) + 4 * off, ptr(arena.grad) + 4 * off,
                                          ptr(arena.state[0]) + 4 * off, None, count,
                                          float(group['lr']), mom, float(group['dampening']),
                                          float(group['weight_decay']), int(group['nesterov']),
                                          0 if inited else 1, int(group['maximize']), st), 'tok_sgd_step')
                if mom != 0 and not inited:
                    for i in range(a, b + 1):
                        self.state[arena.params[i]]['momentum_buffer'] = arena.state_view(0, i)
            self._remember_runs(arena, [(a, b, bool(inited) or mom != 0) for a, b, inited in runs])
        self._repack()
        return loss


class _AdamBase(_ArenaOptimizer):
    _n_state = 2
    _decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0., amsgrad=False, *,
                 foreach=None, maximize: bool = False, capturable: bool = False, differentiable: bool = False,
                 fused=None):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if eps < 0.0:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError(f'Invalid beta parameters: {betas}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if amsgrad or differentiable:
            raise NotImplementedError('amsgrad / differentiable are not supported')
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, maximize=maximize,
                        capturable=bool(capturable))
        super().__init__(params, defaults)

    def _arena_step(self, arena, step_of):
        """Device step counter of a capturable group's arena, created from the (uniform) host-side step of its parameters."""
        # the counter lives ON the arena object: it disappears with the arena when parameters are re-homed (no stale entry
        # under a recycled id()) and is re-seeded from state['step'] after unpickling / load_state_dict
        t = getattr(arena, '_step_dev', None)
        if t is None or t.device != arena.master.device:
            steps = {step_of(i) for i in range(len(arena.params))}
            if len(steps) > 1:
                raise NotImplementedError('capturable Adam: the parameters of a group must share one step count')
            t = torch.full((1,), steps.pop() if steps else 0, dtype=torch.int64, device=arena.master.device)
            arena._step_dev = t
        return t

    def _restore_state(self, old_state):
        for a in self._arenas:
            if a is None:
                continue
            for i, p in enumerate(a.params):
                s = old_state.get(id(p))
                if s and 'exp_avg' in s:
                    a.state_view(0, i).copy_(s['exp_avg'])
                    a.state_view(1, i).copy_(s['exp_avg_sq'])
                    self.state[p].update(step=int(s['step']), exp_avg=a.state_view(0, i),
                                         exp_avg_sq=a.state_view(1, i))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_built()
        lib, st = _C.lib(), stream_ptr()
        for group, arena in zip(self.param_groups, self._arenas):
            if arena is None:
                continue
            b1, b2 = group['betas']

            def step_of(i, arena=arena):
                return int(self.state[arena.params[i]].get('step', 0))
            if group.get('capturable', False):
                # torch.optim.Adam(capturable=True): bias corrections from a device-side step count — nothing the host passes
                # changes from step to step, so the launch sequence can be recorded once (engine/graph.py) and replayed
                tdev = self._arena_step(arena, step_of)
                for a, b, _ in self._runs(arena, lambda i: 0):
                    off = arena.offsets[a]
                    count = arena.padded_end(b) - off
                    _C.check(lib.tok_adam_step_capturable(ptr(arena.master) + 4 * off, ptr(arena.grad) + 4 * off,
                                                          ptr(arena.state[0]) + 4 * off, ptr(arena.state[1]) + 4 * off,
                                                          None, count, float(group['lr']), float(b1), float(b2),
                                                          float(group['eps']), float(group['weight_decay']),
                                                          int(self._decoupled), ptr(tdev), int(group['maximize']), st),
                             'tok_adam_step_capturable')
                    for i in range(a, b + 1):
                        s = self.state[arena.params[i]]
                        if 'exp_avg' not in s:
                            s['exp_avg'] = arena.state_view(0, i)
                            s['exp_avg_sq'] = arena.state_view(1, i)
                        s['step'] = tdev       # (as torch: a device tensor; shared by the parameters of the group)
                _C.check(lib.tok_step_advance(ptr(tdev), st), 'tok_step_advance')
                continue
            runs = self._runs(arena, step_of)
            for a, b, t in runs:
                off = arena.offsets[a]
                count = arena.padded_end(b) - off
                _C.check(lib.tok_adam_step(ptr(arena.master) + 4 * off, ptr(arena.grad) + 4 * off,
                                           ptr(arena.state[0]) + 4 * off, ptr(arena.state[1]) + 4 * off,
                                           None, count, float(group['lr']), float(b1), float(b2),
                                           float(group['eps']), float(group['weight_decay']),
                                           int(self._decoupled), t + 1, int(group['maximize']), st),
                         'tok_adam_step')
                for i in range(a, b + 1):
                    s = self.state[arena.params[i]]
                    if 'exp_avg' not in s:
                        s['exp_avg'] = arena.state_view(0, i)
                        s['exp_avg_sq'] = arena.state_view(1, i)
                    s['step'] = t + 1
            self._remember_runs(arena, [(a, b, t + 1) for a, b, t in runs])
        self._repack()
        return loss


@OPTIMIZERS.register_class
class Adam(_AdamBase):
    _decoupled = False


@OPTIMIZERS.register_class
class AdamW(_AdamBase):
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)


@OPTIMIZERS.register_class
class RMSprop(_ArenaOptimizer):
    """torch.optim.RMSprop semantics (the class the reference registers, optim/optimizers/__init__.py:16) on the flat
    arena: one launch per run of parameters; square_avg / momentum_buffer / grad_avg are arena state slots 0 / 1 / 2."""
    _n_state = 3

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0., momentum=0., centered=False,
                 capturable: bool = False, foreach=None, maximize: bool = False, differentiable: bool = False):
        if lr < 0.0:
            raise ValueError(f'Invalid learning rate: {lr}')
        if eps < 0.0:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if momentum < 0.0:
            raise ValueError(f'Invalid momentum value: {momentum}')
        if weight_decay < 0.0:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if alpha < 0.0:
            raise ValueError(f'Invalid alpha value: {alpha}')
        if differentiable or capturable:
            raise NotImplementedError('differentiable / capturable optimizers are not supported')
        defaults = dict(lr=lr, momentum=momentum, alpha=alpha, eps=eps, centered=centered, weight_decay=weight_decay,
                        maximize=maximize)
        super().__init__(params, defaults)

    _KEYS = ('square_avg', 'momentum_buffer', 'grad_avg')

    def _restore_state(self, old_state):
        for a in self._arenas:
            if a is None:
                continue
            for i, p in enumerate(a.params):
                s = old_state.get(id(p))
                if s and 'square_avg' in s:
                    for slot, key in enumerate(self._KEYS):
                        if s.get(key) is not None:
                            a.state_view(slot, i).copy_(s[key])
                            self.state[p][key] = a.state_view(slot, i)
                    self.state[p]['step'] = int(s.get('step', 0))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._ensure_built()
        lib, st = _C.lib(), stream_ptr()
        for group, arena in zip(self.param_groups, self._arenas):
            if arena is None:
                continue
            for a, b, _ in self._runs(arena, lambda i: 0):
                off = arena.offsets[a]
                count = arena.padded_end(b) - off
                _C.check(lib.tok_rmsprop_step(ptr(arena.master) + 4 * off, ptr(arena.grad) + 4 * off,
                                              ptr(arena.state[0]) + 4 * off, ptr(arena.state[1]) + 4 * off,
                                              ptr(arena.state[2]) + 4 * off, count, float(group['lr']),
                                              float(group['alpha']), float(group['eps']), float(group['weight_decay']),
                                              float(group['momentum']), int(group['centered']), int(group['maximize']),
                                              st), 'tok_rmsprop_step')
                for i in range(a, b + 1):
                    s = self.state[arena.params[i]]
                    if 'square_avg' not in s:
                        s['square_avg'] = arena.state_view(0, i)
                        if group['momentum'] > 0:
                            s['momentum_buffer'] = arena.state_view(1, i)
                        if group['centered']:
                            s['grad_avg'] = arena.state_view(2, i)
                    s['step'] = int(s.get('step', 0)) + 1
        self._repack()
        return loss
