"""`ShapeSpec` (reference detectron2/layers/shape_spec.py): channels/height/width/stride of a feature map."""
from collections import namedtuple


class ShapeSpec(namedtuple("_ShapeSpec", ["channels", "height", "width", "stride"])):
    def __new__(cls, *, channels=None, height=None, width=None, stride=None):
        return super().__new__(cls, channels, height, width, stride)
