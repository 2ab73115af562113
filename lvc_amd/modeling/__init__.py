"""`lvc_amd.modeling`: the reference's registry surface (lvc/modeling/__init__.py) on gfx950 kernels."""
from .anchor_generator import ANCHOR_GENERATOR_REGISTRY, DefaultAnchorGenerator, build_anchor_generator
from .backbone import BACKBONE_REGISTRY, FPN, Backbone, ResNet, build_backbone, build_resnet_backbone, build_resnet_fpn_backbone
from .box_regression import Box2BoxTransform
from .meta_arch import META_ARCH_REGISTRY, GeneralizedRCNN, GeneralizedRCNNRegOnly, ProposalNetwork, build_model
from .poolers import ROIPooler
from .postprocessing import detector_postprocess
from .proposal_generator import PROPOSAL_GENERATOR_REGISTRY, RPN, RPN_HEAD_REGISTRY, StandardRPNHead, build_proposal_generator
from .roi_heads import (ROI_BOX_HEAD_REGISTRY, ROI_HEADS_OUTPUT_REGISTRY, ROI_HEADS_REGISTRY, CascadeROIHeads, ROIHeads, StandardROIHeads,
                        build_box_head, build_roi_heads)

__all__ = [k for k in globals().keys() if not k.startswith("_")]
