from .ddp import GradientAllReducer  # noqa: F401
