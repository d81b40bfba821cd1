from .base import JointLoss  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
from .binary_cross_entropy import BCEWithLogitsLoss  # noqa: F401
from .contrastive import ContrastiveLoss  # noqa: F401
from .dice import DiceLoss  # noqa: F401
from .unsupervised import NT_XentLoss, TripletMarginLoss  # noqa: F401
from .regression import HuberLoss, L1Loss, MSELoss, SmoothL1Loss  # noqa: F401

# torch.nn.Identity is registered as a loss by the reference (losses/__init__.py:35: a task that already returns its loss
# maps it through JointLoss unchanged); it has no arithmetic
from torch.nn import Identity as _Identity  # noqa: E402
from ..constructor import LOSSES as _LOSSES  # noqa: E402

_LOSSES.register_class(_Identity)
