"""LR schedulers: host-side Python that only rewrites ``param_groups[i]['lr']`` — the torch classes
are registered by name as in the reference (``torchok/optim/schedulers/__init__.py:12-30``).  The
six ``timm.scheduler`` classes of the reference are not available offline and are not registered."""
from torch.optim.lr_scheduler import (CosineAnnealingLR, CosineAnnealingWarmRestarts, CyclicLR, ExponentialLR,
                                      LambdaLR, MultiplicativeLR, MultiStepLR, OneCycleLR, ReduceLROnPlateau, StepLR)

from ..constructor import SCHEDULERS

for _cls in (LambdaLR, MultiplicativeLR, StepLR, MultiStepLR, ExponentialLR, CosineAnnealingLR, ReduceLROnPlateau, CyclicLR,
             OneCycleLR, CosineAnnealingWarmRestarts):
    SCHEDULERS.register_class(_cls)
