from .pose_segmentation import pose_segmentation  # noqa: F401
from .generative_functions import generative_model  # noqa: F401
