"""YAML config loading for the hot-path keys of the reference schema
(``torchok/constructor/config_structure.py:7-196``) without hydra / omegaconf (absent on both
boxes): PyYAML (anchors are YAML-native) + a small resolver for the ``${oc.env:X}``,
``${now:fmt}`` and ``${a.b.c}`` interpolations the example configs use
(``examples/configs/classification_cifar10.yaml:46,92-99``), then the schema defaults.
"""
import copy
import datetime
import os
import re
from enum import Enum
from typing import Any, Dict

import yaml


class Phase(Enum):
    TRAIN = 'train'
    VALID = 'valid'
    TEST = 'test'
    PREDICT = 'predict'


class ConfigDict(dict):
    """dict with attribute access (stand-in for omegaconf.DictConfig on the keys we read)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj: Any) -> Any:
    if isinstance(obj, dict):
        return ConfigDict({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


_INTERP = re.compile(r'\$\{([^${}]+)\}')


def _lookup(root: dict, dotted: str):
    cur = root
    for part in dotted.split('.'):
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return cur


def _resolve_str(root: dict, s: str, now: datetime.datetime, depth: int = 0):
    if depth > 16:
        raise ValueError(f'interpolation cycle in {s!r}')

    def sub(m):
        expr = m.group(1).strip()
        if expr.startswith('oc.env:'):
            body = expr[len('oc.env:'):]
            name, _, default = body.partition(',')
            val = os.environ.get(name.strip(), default.strip() if default else None)
            if val is None:
                raise KeyError(f'environment variable {name} is not set')
            return val
        if expr.startswith('now:'):
            return now.strftime(expr[len('now:'):])
        val = _lookup(root, expr)
        if isinstance(val, str):
            val = _resolve_str(root, val, now, depth + 1)
        return str(val)

    whole = _INTERP.fullmatch(s)
    if whole and not whole.group(1).startswith(('oc.env:', 'now:')):
        val = _lookup(root, whole.group(1).strip())   # keep the node type for a pure reference
        return _resolve(root, val, now) if not isinstance(val, str) else _resolve_str(root, val, now, depth + 1)
    prev = None
    while prev != s and _INTERP.search(s):
        prev = s
        s = _INTERP.sub(sub, s)
    return s


def _resolve(root: dict, node: Any, now: datetime.datetime):
    if isinstance(node, dict):
        return {k: _resolve(root, v, now) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(root, v, now) for v in node]
    if isinstance(node, str) and '${' in node:
        return _resolve_str(root, node, now)
    return node


_TASK_DEFAULTS = dict(compute_loss_on_valid=True, params={}, load_checkpoint=None)
_OPT_DEFAULTS = dict(params={}, paramwise_cfg={})
_SCHED_PL_DEFAULTS = dict(interval='epoch', frequency=1, monitor='val_loss', strict=True, name=None)
_LOSS_DEFAULTS = dict(params={}, tag=None, weight=None)
_TOP_DEFAULTS = dict(optimization=None, joint_loss=None, logger=None, metrics=[], callbacks=[],
                     resume_path=None, seed_params=None)
_TOP_KEYS = {'task', 'data', 'trainer', 'optimization', 'joint_loss', 'logger', 'metrics', 'callbacks',
             'resume_path', 'seed_params', 'hydra'}


def apply_schema(cfg: Dict) -> ConfigDict:
    """Fill the defaults of the reference dataclass schema; unknown top-level keys are an error
    (the structured merge of the reference rejects them)."""
    unknown = set(cfg) - _TOP_KEYS
    if unknown:
        raise KeyError(f'unknown config keys: {sorted(unknown)}')
    out = dict(_TOP_DEFAULTS)
    out.update(cfg)
    out.pop('hydra', None)
    if 'task' not in out:
        raise KeyError('config needs a `task` section')
    out['task'] = {**_TASK_DEFAULTS, **out['task']}
    out.setdefault('data', {})
    out.setdefault('trainer', {})
    if out['optimization'] is not None:
        opts = []
        for o in out['optimization']:
            o = dict(o)
            o['optimizer'] = {**_OPT_DEFAULTS, **o['optimizer']}
            sch = o.get('scheduler')
            if sch is not None:
                sch = {'params': {}, **sch}
                sch['pl_params'] = {**_SCHED_PL_DEFAULTS, **(sch.get('pl_params') or {})}
            o['scheduler'] = sch
            opts.append(o)
        out['optimization'] = opts
    if out['joint_loss'] is not None:
        jl = {'normalize_weights': True, **out['joint_loss']}
        jl['losses'] = [{**_LOSS_DEFAULTS, **l} for l in jl['losses']]
        out['joint_loss'] = jl
    return to_config(out)


def load_config(path: str, overrides: Dict[str, Any] = None) -> ConfigDict:
    with open(path) as f:
        raw = yaml.safe_load(f)
    for dotted, value in (overrides or {}).items():
        cur = raw
        parts = dotted.split('.')
        for p in parts[:-1]:
            cur = cur[int(p)] if isinstance(cur, list) else cur.setdefault(p, {})
        cur[parts[-1]] = value
    resolved = _resolve(raw, raw, datetime.datetime.now())
    return apply_schema(resolved)
