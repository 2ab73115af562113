from .boxes import Boxes, BoxMode, pairwise_iou
from .image_list import ImageList
from .instances import Instances

__all__ = ["Boxes", "BoxMode", "pairwise_iou", "ImageList", "Instances"]
