"""PROPOSAL_GENERATOR_REGISTRY + build_proposal_generator (reference
detectron2/modeling/proposal_generator/build.py:4-24): name "PrecomputedProposals" -> None."""
from ...utils.registry import Registry

PROPOSAL_GENERATOR_REGISTRY = Registry("PROPOSAL_GENERATOR")


def build_proposal_generator(cfg, input_shape):
    name = cfg.MODEL.PROPOSAL_GENERATOR.NAME
    if name == "PrecomputedProposals":
        return None
    return PROPOSAL_GENERATOR_REGISTRY.get(name)(cfg, input_shape)
