from .box_head import ROI_BOX_HEAD_REGISTRY, FastRCNNConvFCHead, build_box_head
from .fast_rcnn import ROI_HEADS_OUTPUT_REGISTRY, CosineSimOutputLayers, FastRCNNOutputLayers
from .roi_heads import ROI_HEADS_REGISTRY, ROIHeads, StandardROIHeads, build_roi_heads

__all__ = [k for k in globals().keys() if not k.startswith("_")]
