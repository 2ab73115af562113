"""ROIAlign module (reference detectron2/layers/roi_align.py:22-117) on the gfx950 kernels."""
import torch
from torch import nn
from torch.autograd.function import once_differentiable

from .. import kernels as K


class _ROIAlign(torch.autograd.Function):
    """Forward / backward pair of the reference's `_ROIAlign` (roi_align.py:22-57): the backward is
    once-differentiable and returns a gradient for `input` only."""

    @staticmethod
    def forward(ctx, input, roi, output_size, spatial_scale, sampling_ratio, aligned):
        ctx.save_for_backward(roi)
        ctx.output_size = output_size
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.input_shape = input.size()
        ctx.aligned = aligned
        return K.roi_align_forward(input, roi, spatial_scale, output_size[0], output_size[1], sampling_ratio, aligned)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        bs, ch, h, w = ctx.input_shape
        grad_input = K.roi_align_backward(grad_output, rois, ctx.spatial_scale, ctx.output_size[0],
                                          ctx.output_size[1], bs, ch, h, w, ctx.sampling_ratio, ctx.aligned)
        return grad_input, None, None, None, None, None


class ROIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=True):
        super().__init__()
        self.output_size = output_size if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio
        self.aligned = aligned

    def forward(self, input, rois):
        """input NCHW, rois [K,5] (batch index, x1, y1, x2, y2) -> [K,C,ph,pw]."""
        assert rois.dim() == 2 and rois.size(1) == 5
        return _ROIAlign.apply(input, rois, tuple(self.output_size), self.spatial_scale, self.sampling_ratio,
                               self.aligned)

    def __repr__(self):
        return "ROIAlign(output_size={}, spatial_scale={}, sampling_ratio={}, aligned={})".format(
            self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, aligned):
    return ROIAlign(output_size, spatial_scale, sampling_ratio, aligned)(input, rois)
