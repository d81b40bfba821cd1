from .base import JointLoss  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
