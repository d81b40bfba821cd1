"""Name -> callable registries with the semantics of the reference's
``torchok/constructor/registry.py:10-138`` (own implementation; no timm import).

Contract kept: ``REG.get(name)`` / ``REG[name]`` raise ``KeyError`` for a missing entry
(reference ``registry.py:57-58``), ``register_class`` raises ``TypeError`` for a non-callable and
``KeyError`` for a duplicate name (``:76-81``), appends the name to the defining module's
``__all__`` (``:87-92``), and ``list_models`` filters with fnmatch and sorts in natural order
(``:101-138``).
"""
import fnmatch
import re
import sys
from collections import defaultdict
from typing import Callable, List, Union


def _natural_key(string_: str):
    # [timm 0.6.13] timm.models.registry._natural_key
    return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', string_.lower())]


class Registry:
    def __init__(self, name: str):
        self.name = name
        self.entrypoints = {}
        self.module_to_objects = defaultdict(set)
        self.object_to_module = {}

    def __repr__(self):
        return f'{self.__class__.__name__}(name={self.name}, items={list(self.entrypoints)})'

    def __contains__(self, item):
        return item in self.entrypoints

    def __getitem__(self, key):
        return self.get(key)

    def get(self, key: str):
        if key not in self.entrypoints:
            raise KeyError(f'{key} is not in the {self.name} registry')
        return self.entrypoints[key]

    def register_class(self, fn: Callable):
        if not callable(fn):
            raise TypeError(f'{fn} must be callable')
        name = fn.__name__
        if name in self.entrypoints:
            raise KeyError(f'{name} is already registered in {self.name}')
        mod = sys.modules.get(fn.__module__)
        parts = fn.__module__.split('.')
        module_name = parts[-1] if parts else ''
        if mod is not None:
            if hasattr(mod, '__all__'):
                mod.__all__.append(name)
            else:
                mod.__all__ = [name]
        self.entrypoints[name] = fn
        self.object_to_module[name] = module_name
        self.module_to_objects[module_name].add(name)
        return fn

    def list_models(self, filter: str = '', module: str = '',
                    exclude_filters: Union[str, List[str]] = '') -> List[str]:
        names = list(self.module_to_objects[module]) if module else list(self.entrypoints.keys())
        if filter:
            selected = set()
            for f in (filter if isinstance(filter, (tuple, list)) else [filter]):
                selected.update(fnmatch.filter(names, f))
            names = selected
        if exclude_filters:
            if not isinstance(exclude_filters, (tuple, list)):
                exclude_filters = [exclude_filters]
            for xf in exclude_filters:
                names = set(names).difference(fnmatch.filter(names, xf))
        return list(sorted(names, key=_natural_key))
