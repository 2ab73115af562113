from .build import PROPOSAL_GENERATOR_REGISTRY, build_proposal_generator
from .rbg import RBG
from .rpn import RPN, RPN_HEAD_REGISTRY, StandardRPNHead, build_rpn_head

__all__ = [k for k in globals().keys() if not k.startswith("_")]
