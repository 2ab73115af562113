from .build import META_ARCH_REGISTRY, build_model
from .rcnn import GeneralizedRCNN, GeneralizedRCNNRegOnly, ProposalNetwork

__all__ = [k for k in globals().keys() if not k.startswith("_")]
