"""Backbone base class and registry (reference detectron2/modeling/backbone/backbone.py:10-53,
build.py:7-33).  `forward(NCHW) -> dict[str, NCHW-shaped tensor]`, `output_shape()`, `size_divisibility`."""
from torch import nn

from ...layers import ShapeSpec
from ...utils.registry import Registry

BACKBONE_REGISTRY = Registry("BACKBONE")


class Backbone(nn.Module):
    def __init__(self):
        super().__init__()

    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {
            name: ShapeSpec(channels=self._out_feature_channels[name], stride=self._out_feature_strides[name])
            for name in self._out_features
        }


def build_backbone(cfg, input_shape=None):
    if input_shape is None:
        input_shape = ShapeSpec(channels=len(cfg.MODEL.PIXEL_MEAN))
    backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, input_shape)
    assert isinstance(backbone, Backbone)
    return backbone
