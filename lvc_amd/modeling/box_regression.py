"""Box2BoxTransform (reference detectron2/modeling/box_regression.py:15-110): the (dx,dy,dw,dh)
parameterisation.  On the inference hot path the decode runs inside csrc/boxes.hip (`apply_deltas`
device function, same association); the methods here serve callers and the training-time targets."""
import math

import torch

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp=_DEFAULT_SCALE_CLAMP):
        self.weights = weights
        self.scale_clamp = scale_clamp

    def get_deltas(self, src_boxes, target_boxes):
        assert isinstance(src_boxes, torch.Tensor), type(src_boxes)
        assert isinstance(target_boxes, torch.Tensor), type(target_boxes)
        sw = src_boxes[:, 2] - src_boxes[:, 0]
        sh = src_boxes[:, 3] - src_boxes[:, 1]
        scx = src_boxes[:, 0] + 0.5 * sw
        scy = src_boxes[:, 1] + 0.5 * sh
        tw = target_boxes[:, 2] - target_boxes[:, 0]
        th = target_boxes[:, 3] - target_boxes[:, 1]
        tcx = target_boxes[:, 0] + 0.5 * tw
        tcy = target_boxes[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        deltas = torch.stack((wx * (tcx - scx) / sw, wy * (tcy - scy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)
        assert (sw > 0).all().item(), "Input boxes to Box2BoxTransform are not valid!"
        return deltas

    def apply_deltas(self, deltas, boxes):
        boxes = boxes.to(deltas.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = deltas[:, 0::4] / wx
        dy = deltas[:, 1::4] / wy
        dw = torch.clamp(deltas[:, 2::4] / ww, max=self.scale_clamp)
        dh = torch.clamp(deltas[:, 3::4] / wh, max=self.scale_clamp)
        pcx = dx * widths[:, None] + ctr_x[:, None]
        pcy = dy * heights[:, None] + ctr_y[:, None]
        pw = torch.exp(dw) * widths[:, None]
        ph = torch.exp(dh) * heights[:, None]
        out = torch.zeros_like(deltas)
        out[:, 0::4] = pcx - 0.5 * pw
        out[:, 1::4] = pcy - 0.5 * ph
        out[:, 2::4] = pcx + 0.5 * pw
        out[:, 3::4] = pcy + 0.5 * ph
        return out
