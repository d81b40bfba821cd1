from .hrnet_classification import HRNetClassificationNeck  # noqa: F401
from .hrnet_segmentation import HRNetSegmentationNeck  # noqa: F401
