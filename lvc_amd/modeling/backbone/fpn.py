"""FPN on top of the ResNet trunk (reference detectron2/modeling/backbone/fpn.py:16-152, 165-177,
201-221): same module names (`fpn_lateral{2..5}`, `fpn_output{2..5}`, `bottom_up`, `top_block`).

Launch plan per level (top-down): the lateral 1x1 conv's epilogue adds the nearest-x2-upsampled
coarser map (reference fpn.py:131-133 does interpolate + add as two extra passes over HBM), then the
3x3 output conv; p6 is a stride-2 subsample of p5.
"""
import math

import torch
from torch import nn

from ... import kernels as K
from ...layers import Conv2d, ShapeSpec, get_norm
from ...layers.layout import require_device, to_nchw_view
from ...utils import weight_init
from .backbone import BACKBONE_REGISTRY, Backbone
from .resnet import _as_nhwc4, build_resnet_backbone


class _Subsample2(torch.autograd.Function):
    """max_pool2d(kernel 1, stride 2) = every other pixel; backward scatters the gradient back to those pixels."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = (x.shape[1], x.shape[2])
        return K.maxpool2d_nhwc(x, 1, 2, 0)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return K.scatter_stride2(g, ctx.hw[0], ctx.hw[1])


class LastLevelMaxPool(nn.Module):
    """p6 = max_pool2d(p5, kernel_size=1, stride=2) (reference fpn.py:165-177)."""

    def __init__(self):
        super().__init__()
        self.num_levels = 1
        self.in_feature = "p5"

    def forward_nhwc(self, x):
        if torch.is_grad_enabled() and x.requires_grad:
            return [_Subsample2.apply(x)]
        return [K.maxpool2d_nhwc(x, 1, 2, 0)]

    def forward(self, x):
        from ...layers.layout import to_nhwc

        return [to_nchw_view(t) for t in self.forward_nhwc(to_nhwc(x))]


MERGE_OUTPUT_CONVS = True      # the 3x3 output convs of all levels as one launch (FPN.forward_nhwc)


class FPN(Backbone):
    def __init__(self, bottom_up, in_features, out_channels, norm="", top_block=None, fuse_type="sum"):
        super().__init__()
        assert isinstance(bottom_up, Backbone)
        assert fuse_type in {"avg", "sum"}       # reference fpn.py:63
        # "avg" (reference fpn.py:133-134; no shipped config selects it): the fused lateral + upsample-add launch, then an exact halving
        self._fuse_type = fuse_type
        input_shapes = bottom_up.output_shape()
        strides = [input_shapes[f].stride for f in in_features]
        in_channels_per_feature = [input_shapes[f].channels for f in in_features]
        for i in range(1, len(strides)):
            assert strides[i] == 2 * strides[i - 1], "Strides {} {} are not log2 contiguous".format(strides[i], strides[i - 1])
        lateral_convs, output_convs = [], []
        use_bias = norm == ""
        for idx, in_channels in enumerate(in_channels_per_feature):
            lateral_conv = Conv2d(in_channels, out_channels, kernel_size=1, bias=use_bias, norm=get_norm(norm, out_channels))
            output_conv = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=use_bias,
                                 norm=get_norm(norm, out_channels))
            weight_init.c2_xavier_fill(lateral_conv)
            weight_init.c2_xavier_fill(output_conv)
            stage = int(math.log2(strides[idx]))
            self.add_module("fpn_lateral{}".format(stage), lateral_conv)
            self.add_module("fpn_output{}".format(stage), output_conv)
            lateral_convs.append(lateral_conv)
            output_convs.append(output_conv)
        # plain lists (not ModuleLists): registered above under the reference's names
        self.lateral_convs = lateral_convs[::-1]
        self.output_convs = output_convs[::-1]
        self.top_block = top_block
        self.in_features = in_features
        self.bottom_up = bottom_up
        self._out_feature_strides = {"p{}".format(int(math.log2(s))): s for s in strides}
        if self.top_block is not None:
            for s in range(stage, stage + self.top_block.num_levels):
                self._out_feature_strides["p{}".format(s + 1)] = 2 ** (s + 1)
        self._out_features = list(self._out_feature_strides.keys())
        self._out_feature_channels = {k: out_channels for k in self._out_features}
        self._size_divisibility = strides[-1]

    @property
    def size_divisibility(self):
        return self._size_divisibility

    def _fuse(self, summed):
        return summed * 0.5 if self._fuse_type == "avg" else summed

    def forward_nhwc(self, x4):
        bottom_up = self.bottom_up.forward_nhwc(x4)
        x = [bottom_up[f] for f in self.in_features[::-1]]
        results = []
        grad_free = not torch.is_grad_enabled() or not any(p.requires_grad for p in self.parameters())
        if grad_free and MERGE_OUTPUT_CONVS and all(c.norm is None and c.activation is None for c in self.output_convs):
            # the top-down path first (lateral + upsample-add, coarse to fine), then the output convs of ALL levels as one launch
            # (kernels.conv3x3_levels: a layer per map; the small levels no longer pay a launch each that cannot fill the chip)
            prevs = [self.lateral_convs[0].forward_nhwc(x[0])]
            for feat, lateral in zip(x[1:], self.lateral_convs[1:]):
                prevs.append(self._fuse(lateral.forward_nhwc(feat, residual=prevs[-1], res_mode=2)))
            order = list(range(len(prevs)))[::-1]          # finest (largest) map first
            outs = K.conv3x3_levels([prevs[i] for i in order], [self.output_convs[i].packed() for i in order], relu=False)
            results = list(outs)                           # fine to coarse, the order of self._out_features
        else:
            prev = self.lateral_convs[0].forward_nhwc(x[0])
            results.append(self.output_convs[0].forward_nhwc(prev))
            for feat, lateral, output in zip(x[1:], self.lateral_convs[1:], self.output_convs[1:]):
                prev = self._fuse(lateral.forward_nhwc(feat, residual=prev, res_mode=2))  # lateral + upsample(prev)
                results.insert(0, output.forward_nhwc(prev))
        if self.top_block is not None:
            src = bottom_up.get(self.top_block.in_feature, None)
            if src is None:
                src = results[self._out_features.index(self.top_block.in_feature)]
            results.extend(self.top_block.forward_nhwc(src))
        assert len(self._out_features) == len(results)
        return dict(zip(self._out_features, results))

    def forward(self, x):
        require_device(x, "FPN")
        return {k: to_nchw_view(v) for k, v in self.forward_nhwc(_as_nhwc4(x)).items()}


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, input_shape: ShapeSpec):
    bottom_up = build_resnet_backbone(cfg, input_shape)
    return FPN(bottom_up=bottom_up, in_features=cfg.MODEL.FPN.IN_FEATURES, out_channels=cfg.MODEL.FPN.OUT_CHANNELS,
               norm=cfg.MODEL.FPN.NORM, top_block=LastLevelMaxPool(), fuse_type=cfg.MODEL.FPN.FUSE_TYPE)
