"""Programmatic equivalents of the shipped yaml files, for environments where the reference's `configs/`
tree is not present (the GPU box, bench.py).  Values are the ones `configs/Base-RCNN-FPN.yaml:1-38` sets on
top of the defaults; loading the real yaml through `cfg.merge_from_file` gives the same node (asserted by
tests/test_config_dropin.py whenever /root/reference exists)."""
from . import get_cfg


def base_rcnn_fpn(depth=50, num_classes=80, device="cuda"):
    cfg = get_cfg()
    M = cfg.MODEL
    M.META_ARCHITECTURE = "GeneralizedRCNN"
    M.DEVICE = device
    M.MASK_ON = False
    M.BACKBONE.NAME = "build_resnet_fpn_backbone"
    M.RESNETS.OUT_FEATURES = ["res2", "res3", "res4", "res5"]
    M.RESNETS.DEPTH = depth
    M.FPN.IN_FEATURES = ["res2", "res3", "res4", "res5"]
    M.ANCHOR_GENERATOR.SIZES = [[32], [64], [128], [256], [512]]
    M.ANCHOR_GENERATOR.ASPECT_RATIOS = [[0.5, 1.0, 2.0]]
    M.RPN.IN_FEATURES = ["p2", "p3", "p4", "p5", "p6"]
    M.RPN.PRE_NMS_TOPK_TRAIN = 2000
    M.RPN.PRE_NMS_TOPK_TEST = 1000
    M.RPN.POST_NMS_TOPK_TRAIN = 1000
    M.RPN.POST_NMS_TOPK_TEST = 1000
    M.ROI_HEADS.NAME = "StandardROIHeads"
    M.ROI_HEADS.IN_FEATURES = ["p2", "p3", "p4", "p5"]
    M.ROI_HEADS.NUM_CLASSES = num_classes
    M.ROI_BOX_HEAD.NAME = "FastRCNNConvFCHead"
    M.ROI_BOX_HEAD.NUM_FC = 2
    M.ROI_BOX_HEAD.POOLER_RESOLUTION = 7
    return cfg
