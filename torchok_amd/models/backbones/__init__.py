from . import resnet  # noqa: F401
