from .box_head import ROI_BOX_HEAD_REGISTRY, FastRCNNConvFCHead, build_box_head
from .fast_rcnn import (ROI_HEADS_OUTPUT_REGISTRY, BoxOnlyLayers, BoxOnlyLayersCascade, CosineSimOutputLayers,
                        FastRCNNOutputLayers)
from .roi_heads import ROI_HEADS_REGISTRY, ROIHeads, StandardROIHeads, build_roi_heads
from .cascade_rcnn import CascadeROIHeads  # noqa: E402

__all__ = [k for k in globals().keys() if not k.startswith("_")]
