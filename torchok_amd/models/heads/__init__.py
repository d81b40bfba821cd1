from .linear_head import ClassificationHead, LinearHead  # noqa: F401
from .arcface_head import ArcFaceHead  # noqa: F401
from .segmentation_head import SegmentationHead  # noqa: F401
from .ocr_head import OCRSegmentationHead  # noqa: F401
