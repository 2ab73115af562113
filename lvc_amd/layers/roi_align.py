"""ROIAlign module (reference detectron2/layers/roi_align.py:63-117) on the gfx950 kernel."""
from torch import nn

from .. import kernels as K


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
        return K.roi_align_forward(input, rois, self.spatial_scale, self.output_size[0], self.output_size[1],
                                   self.sampling_ratio, self.aligned)

    def __repr__(self):
        return "ROIAlign(output_size={}, spatial_scale={}, sampling_ratio={}, aligned={})".format(
            self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio, aligned):
    return ROIAlign(output_size, spatial_scale, sampling_ratio, aligned)(input, rois)
