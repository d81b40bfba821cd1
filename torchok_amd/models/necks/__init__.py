from .hrnet_segmentation import HRNetSegmentationNeck  # noqa: F401
