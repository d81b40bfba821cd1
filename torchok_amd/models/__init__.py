from . import backbones, heads, necks, poolings  # noqa: F401
