"""ROIAlign module (reference detectron2/layers/roi_align.py:22-117) on the gfx950 kernels."""
import torch
from torch import nn

from .. import ops


def _roi_align(input, roi, output_size, spatial_scale, sampling_ratio, aligned):
    """The reference's `_ROIAlign.apply` (roi_align.py:22-57) = the custom op `torch.ops.lvc_amd.roi_align_forward`, whose
    registered autograd formula calls `torch.ops.lvc_amd.roi_align_backward` (gradient for `input` only)."""
    return ops.roi_align_forward(input, roi, float(spatial_scale), int(output_size[0]), int(output_size[1]),
                                 int(sampling_ratio), bool(aligned))


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
        return _roi_align(input, rois, tuple(self.output_size), self.spatial_scale, self.sampling_ratio, self.aligned)

    def __repr__(self):
        return "ROIAlign(output_size={}, spatial_scale={}, sampling_ratio={}, aligned={})".format(
            self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, aligned):
    return ROIAlign(output_size, spatial_scale, sampling_ratio, aligned)(input, rois)
