"""DefaultAnchorGenerator (reference detectron2/modeling/anchor_generator.py:37-49, 79-223).

Owns the `cell_anchors.{i}` buffers (part of the state_dict contract).  The full per-location anchor
grid (268 569 x 4 floats for 800x1344) is never materialised on the hot path: the RPN decode kernel
rebuilds `shift + cell_anchor` for just the top-k candidates (csrc/boxes.hip).  `forward()` still
returns the full grid as `Boxes` for callers that want it (training-time matching).
"""
import math

import torch
from torch import nn

from ..structures import Boxes
from ..utils.registry import Registry

ANCHOR_GENERATOR_REGISTRY = Registry("ANCHOR_GENERATOR")


class BufferList(nn.Module):
    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b)

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


def _broadcast_params(params, num_features, name):
    assert isinstance(params, (list, tuple)), "{} in anchor generator has to be a list! Got {}.".format(name, params)
    assert len(params), "{} in anchor generator cannot be empty!".format(name)
    if not isinstance(params[0], (list, tuple)):
        return [params] * num_features
    if len(params) == 1:
        return list(params) * num_features
    assert len(params) == num_features, (
        "Got {} of length {} in anchor generator, but the number of input features is {}!".format(name, len(params), num_features))
    return params


@ANCHOR_GENERATOR_REGISTRY.register()
class DefaultAnchorGenerator(nn.Module):
    box_dim = 4

    def __init__(self, cfg=None, input_shape=None, *, sizes=None, aspect_ratios=None, strides=None, offset=None):
        super().__init__()
        if cfg is not None:
            sizes = cfg.MODEL.ANCHOR_GENERATOR.SIZES
            aspect_ratios = cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS
            strides = [x.stride for x in input_shape]
            offset = cfg.MODEL.ANCHOR_GENERATOR.OFFSET
        self.strides = strides
        self.num_features = len(self.strides)
        sizes = _broadcast_params(sizes, self.num_features, "sizes")
        aspect_ratios = _broadcast_params(aspect_ratios, self.num_features, "aspect_ratios")
        self.cell_anchors = BufferList([self.generate_cell_anchors(s, a).float() for s, a in zip(sizes, aspect_ratios)])
        self.offset = offset
        assert 0.0 <= self.offset < 1.0, self.offset

    @property
    def num_anchors(self):
        return [len(c) for c in self.cell_anchors]

    num_cell_anchors = num_anchors

    @staticmethod
    def generate_cell_anchors(sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
        anchors = []
        for size in sizes:
            area = size ** 2.0
            for ar in aspect_ratios:
                w = math.sqrt(area / ar)
                h = ar * w
                anchors.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(anchors)

    def _grid_anchors(self, grid_sizes):
        out = []
        for (gh, gw), stride, base in zip(grid_sizes, self.strides, self.cell_anchors):
            sx = torch.arange(self.offset * stride, gw * stride, step=stride, dtype=torch.float32, device=base.device)
            sy = torch.arange(self.offset * stride, gh * stride, step=stride, dtype=torch.float32, device=base.device)
            yy, xx = torch.meshgrid(sy, sx, indexing="ij")
            xx, yy = xx.reshape(-1), yy.reshape(-1)
            shifts = torch.stack((xx, yy, xx, yy), dim=1)
            out.append((shifts.view(-1, 1, 4) + base.view(1, -1, 4)).reshape(-1, 4))
        return out

    def forward(self, features):
        grid_sizes = [f.shape[-2:] for f in features]
        return [Boxes(x) for x in self._grid_anchors(grid_sizes)]


def build_anchor_generator(cfg, input_shape):
    return ANCHOR_GENERATOR_REGISTRY.get(cfg.MODEL.ANCHOR_GENERATOR.NAME)(cfg, input_shape)
