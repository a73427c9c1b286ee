from .pose_segmentation import pose_segmentation  # noqa: F401
