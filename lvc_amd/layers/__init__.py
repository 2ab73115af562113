from .batch_norm import FrozenBatchNorm2d, get_norm
from .nms import batched_nms, nms
from .roi_align import ROIAlign, roi_align
from .shape_spec import ShapeSpec
from .wrappers import Conv2d, Linear, cat, nonzero_tuple

__all__ = [k for k in globals().keys() if not k.startswith("_")]
