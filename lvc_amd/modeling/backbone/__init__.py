from .backbone import BACKBONE_REGISTRY, Backbone, build_backbone
from .fpn import FPN, LastLevelMaxPool, build_resnet_fpn_backbone
from .resnet import BasicStem, BottleneckBlock, ResNet, build_resnet_backbone

__all__ = [k for k in globals().keys() if not k.startswith("_")]
