"""META_ARCH_REGISTRY + build_model (reference lvc/modeling/meta_arch/build.py:3-17)."""
from ...utils.registry import Registry

META_ARCH_REGISTRY = Registry("META_ARCH")


def build_model(cfg):
    return META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
