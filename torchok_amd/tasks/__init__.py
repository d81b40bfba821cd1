from .base import BaseTask  # noqa: F401
from .classification import ClassificationTask  # noqa: F401
from .multihead_classification import MultiHeadClassificationTask  # noqa: F401
from .pairwise_task import PairwiseLearnTask  # noqa: F401
from .segmentation import SegmentationTask  # noqa: F401
from .unsupervised import SimCLRTask, TripletLearnTask  # noqa: F401
