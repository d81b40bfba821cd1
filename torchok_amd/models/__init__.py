from . import backbones, heads, poolings  # noqa: F401
