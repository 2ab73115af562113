"""ROIPooler (reference detectron2/modeling/poolers.py:99-246) for ROIAlign / ROIAlignV2.

MI355X-first: instead of the reference's per-level gather -> ROIAlign -> scatter loop (4 launches of
3 ops each and a `nonzero` host sync per level), level assignment is one small kernel and ROIAlign
over ALL levels is one launch (each RoI's workgroup picks its level's feature pointer); the output
is written once, in proposal order.
"""
import math

import torch
from torch import nn

from .. import kernels as K
from ..layers.layout import to_nchw_view, to_nhwc


class _PoolAllLevels(torch.autograd.Function):
    """ROIAlign over all pyramid levels in one launch, differentiable w.r.t. the level maps (the scatter of
    `lvc_roi_align_fpn_backward_nhwc`); boxes get no gradient, as in the reference's `_ROIAlign` (roi_align.py:22-57)."""

    @staticmethod
    def forward(ctx, pooler, boxes, status, *feats):
        levels, rois = K.assign_levels_rois(boxes, pooler.min_level, pooler.max_level, pooler.canonical_box_size,
                                            pooler.canonical_level)
        if len(feats) == 1:
            levels = None
        ctx.pooler, ctx.rois, ctx.levels = pooler, rois, levels
        ctx.shapes = [tuple(f.shape) for f in feats]
        return K.roi_align_fpn_nhwc(list(feats), pooler.scales, rois, levels, pooler.output_size[0], pooler.output_size[1],
                                    pooler.sampling_ratio, pooler.aligned, status=status)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        pl = ctx.pooler
        grads = K.roi_align_fpn_backward_nhwc(grad, ctx.shapes, pl.scales, ctx.rois, ctx.levels, pl.sampling_ratio, pl.aligned)
        return (None, None, None) + tuple(grads)


class ROIPooler(nn.Module):
    def __init__(self, output_size, scales, sampling_ratio, pooler_type, canonical_box_size=224, canonical_level=4):
        super().__init__()
        if isinstance(output_size, int):
            output_size = (output_size, output_size)
        assert len(output_size) == 2 and isinstance(output_size[0], int) and isinstance(output_size[1], int)
        self.output_size = output_size
        if pooler_type == "ROIAlign":
            self.aligned = False
        elif pooler_type == "ROIAlignV2":
            self.aligned = True
        else:
            raise NotImplementedError("pooler type '{}' is not used by the shipped configs".format(pooler_type))
        self.scales = tuple(scales)
        self.sampling_ratio = sampling_ratio
        min_level = -(math.log2(scales[0]))
        max_level = -(math.log2(scales[-1]))
        assert math.isclose(min_level, int(min_level)) and math.isclose(max_level, int(max_level)), \
            "Featuremap stride is not power of 2!"
        self.min_level, self.max_level = int(min_level), int(max_level)
        assert len(scales) == self.max_level - self.min_level + 1, "[ROIPooler] Sizes of input featuremaps do not form a pyramid!"
        assert 0 <= self.min_level <= self.max_level
        self.canonical_level = canonical_level
        assert canonical_box_size > 0
        self.canonical_box_size = canonical_box_size

    def pool_nhwc(self, feats_nhwc, boxes, status=None):
        """feats_nhwc: list of [B,H,W,C]; boxes: [B,R,4] device tensor (zero rows = padding).
        Returns [B*R, ph, pw, C] (channels-last rows, the layout the box-head GEMM consumes)."""
        if torch.is_grad_enabled() and any(f.requires_grad for f in feats_nhwc):
            return _PoolAllLevels.apply(self, boxes.detach(), status, *feats_nhwc)
        levels, rois = K.assign_levels_rois(boxes, self.min_level, self.max_level, self.canonical_box_size,
                                            self.canonical_level)
        if len(feats_nhwc) == 1:
            levels = None
        return K.roi_align_fpn_nhwc(feats_nhwc, self.scales, rois, levels, self.output_size[0], self.output_size[1],
                                    self.sampling_ratio, self.aligned, status=status)

    def forward(self, x, box_lists):
        """Reference signature: x list of NCHW maps, box_lists list[Boxes] -> [M, C, ph, pw]."""
        assert isinstance(x, list) and isinstance(box_lists, list), "Arguments to pooler must be lists"
        assert len(x) == len(self.scales)
        assert len(box_lists) == x[0].size(0)
        counts = [len(b) for b in box_lists]
        M = sum(counts)
        C = x[0].shape[1]
        if M == 0:
            return torch.zeros((0, C) + self.output_size, device=x[0].device, dtype=x[0].dtype)
        R = max(counts)
        boxes = torch.zeros(len(box_lists), R, 4, device=x[0].device)
        for i, b in enumerate(box_lists):
            boxes[i, : counts[i]] = b.tensor
        out = self.pool_nhwc([to_nhwc(f) for f in x], boxes)
        out = out.view(len(box_lists), R, self.output_size[0], self.output_size[1], C)
        out = torch.cat([out[i, : counts[i]] for i in range(len(box_lists))], 0)
        return out.permute(0, 3, 1, 2)  # NCHW-shaped, channels-last storage
