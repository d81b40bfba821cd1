"""torchok_amd — the MI355X-native vision-training hot path behind the TorchOk registry / Task API.

    from torchok_amd.constructor import BACKBONES, HEADS, POOLINGS, LOSSES, OPTIMIZERS, TASKS
    task = TASKS.get('ClassificationTask')(cfg, **cfg.task.params).cuda()

Everything numeric runs in hand-written HIP kernels for gfx950 behind the C ABI of include/tok.h
(torchok_amd/lib/libtok_gfx950.so, built by __graft_entry__.build()).  No CPU fallback exists.
"""
from . import constructor  # noqa: F401
from . import models, losses, metrics, retrieval, optim, tasks  # noqa: F401
from .constructor import (BACKBONES, HEADS, LOSSES, METRICS, NECKS, OPTIMIZERS, POOLINGS, SCHEDULERS, TASKS)  # noqa: F401
from .constructor.config import load_config  # noqa: F401

__version__ = '0.1.0'
