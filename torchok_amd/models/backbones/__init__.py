from . import davit, hrnet, resnet, swin  # noqa: F401
