from .pooling import Pooling, PoolingLinear  # noqa: F401
