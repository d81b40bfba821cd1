from . import hrnet, resnet  # noqa: F401
