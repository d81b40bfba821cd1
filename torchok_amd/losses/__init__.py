from .base import JointLoss  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
from .binary_cross_entropy import BCEWithLogitsLoss  # noqa: F401
from .contrastive import ContrastiveLoss  # noqa: F401
from .dice import DiceLoss  # noqa: F401
from .unsupervised import NT_XentLoss, TripletMarginLoss  # noqa: F401
from .regression import HuberLoss, L1Loss, MSELoss, SmoothL1Loss  # noqa: F401
