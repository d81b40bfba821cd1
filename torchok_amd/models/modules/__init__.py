from .convbnact import ConvBnAct  # noqa: F401
