from . import hrnet, resnet, swin  # noqa: F401
