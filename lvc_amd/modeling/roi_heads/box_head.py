"""FastRCNNConvFCHead (reference lvc/modeling/roi_heads/box_head.py:18-95): NUM_CONV 3x3 convs then
NUM_FC fully connected layers with ReLU (+ dropout, identity in eval).  Parameter names `conv{k}`,
`fc{k}`.  The first fc consumes ROIAlign's channels-last [M,7,7,C] rows directly: its packed weight is
column-permuted from the reference's (c,h,w) flatten order to (h,w,c) once at pack time, so no
transpose of the 50 MB pooled tensor is ever made."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import kernels as K
from ...layers import Conv2d, Linear, ShapeSpec, get_norm
from ...layers.wrappers import _PackedCache, linear_fn
from ...utils import weight_init
from ...utils.registry import Registry

ROI_BOX_HEAD_REGISTRY = Registry("ROI_BOX_HEAD")


@ROI_BOX_HEAD_REGISTRY.register()
class FastRCNNConvFCHead(nn.Module):
    def __init__(self, cfg, input_shape: ShapeSpec):
        super().__init__()
        H = cfg.MODEL.ROI_BOX_HEAD
        num_conv, conv_dim, num_fc, fc_dim, norm = H.NUM_CONV, H.CONV_DIM, H.NUM_FC, H.FC_DIM, H.NORM
        dropout = H.DROPOUT
        assert num_conv + num_fc > 0
        self._output_size = (input_shape.channels, input_shape.height, input_shape.width)
        self.conv_norm_relus = []
        for k in range(num_conv):
            conv = Conv2d(self._output_size[0], conv_dim, kernel_size=3, padding=1, bias=not norm,
                          norm=get_norm(norm, conv_dim), activation=F.relu)
            self.add_module("conv{}".format(k + 1), conv)
            self.conv_norm_relus.append(conv)
            self._output_size = (conv_dim, self._output_size[1], self._output_size[2])
        self.dropout = nn.Dropout(p=dropout) if dropout else nn.Identity()
        self.fcs = []
        self._fc1_chw = self._output_size if num_fc else None
        for k in range(num_fc):
            fc = Linear(int(np.prod(self._output_size)), fc_dim)
            self.add_module("fc{}".format(k + 1), fc)
            self.fcs.append(fc)
            self._output_size = fc_dim
        for layer in self.conv_norm_relus:
            weight_init.c2_msra_fill(layer)
        for layer in self.fcs:
            weight_init.c2_xavier_fill(layer)
        self._fc1_hwc = _PackedCache()

    def _packed_fc1_hwc(self):
        fc = self.fcs[0]
        C, Hh, Ww = self._fc1_chw

        def build():
            w = fc.weight.view(fc.out_features, C, Hh, Ww).permute(0, 2, 3, 1).reshape(fc.out_features, -1)
            return K.pack_linear(w, fc.bias)

        return self._fc1_hwc.get([fc.weight, fc.bias], build)

    def forward_nhwc(self, x):
        """x: [M, h, w, C] channels-last pooled features."""
        # dropout (reference box_head.py:88-89, after every FC's ReLU; identity in eval): torch's own nn.Dropout on the
        # device tensor -- its mask comes from torch's generator exactly as the reference's does
        for layer in self.conv_norm_relus:
            x = layer.forward_nhwc(x)
        if len(self.fcs):
            M = x.shape[0]
            fc = self.fcs[0]
            C, Hh, Ww = self._fc1_chw
            x = linear_fn(
                x.reshape(M, -1), fc.weight, fc.bias, self._packed_fc1_hwc(), relu=True,
                w_packed_layout=lambda: fc.weight.detach().view(fc.out_features, C, Hh, Ww).permute(0, 2, 3, 1).reshape(fc.out_features, -1),
                w_view=lambda dw: dw.view(fc.out_features, Hh, Ww, C).permute(0, 3, 1, 2).reshape(fc.out_features, -1))
            x = self.dropout(x)
            for fc in self.fcs[1:]:
                x = self.dropout(fc(x, relu=True))
        return x

    def forward(self, x):
        """Reference signature: x [M,C,h,w] (any memory format) -> [M, fc_dim]."""
        if x.dim() == 4:
            from ...layers.layout import to_nhwc

            return self.forward_nhwc(to_nhwc(x))
        for fc in self.fcs:
            x = self.dropout(fc(x, relu=True))
        return x

    @property
    def output_size(self):
        return self._output_size


def build_box_head(cfg, input_shape):
    return ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, input_shape)
