from .linear_head import ClassificationHead, LinearHead  # noqa: F401
