from . import optimizers, schedulers  # noqa: F401
