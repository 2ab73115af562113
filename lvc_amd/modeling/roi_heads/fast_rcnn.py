"""Fast R-CNN output layers + inference (reference lvc/modeling/roi_heads/fast_rcnn.py):
`FastRCNNOutputLayers` :499-598, `CosineSimOutputLayers` :721-841, `fast_rcnn_inference` :51-137,
`FastRCNNOutputs.predict_boxes/predict_probs/inference` :440-493.

Parameter names `cls_score.{weight,bias}`, `bbox_pred.{weight,bias}` (no cls bias for the cosine
layer).  Inference = one fused GEMM for both linears (two for the cosine variant, whose inputs
differ) + lvc_fast_rcnn_inference (softmax, per-class decode, clip, threshold, per-class NMS, top-k,
optional detector_postprocess) on device.
"""
import torch
from torch import nn

from ... import kernels as K
from ...layers import Linear, ShapeSpec
from ...layers.wrappers import _PackedCache
from ...utils.registry import Registry
from ..box_regression import Box2BoxTransform

ROI_HEADS_OUTPUT_REGISTRY = Registry("ROI_HEADS_OUTPUT")


class _OutputLayersBase(nn.Module):
    def _common(self, cfg, input_shape):
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        self.input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.cls_agnostic_bbox_reg = cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG
        self.box2box_transform = Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.smooth_l1_beta = cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.box_reg_loss_type = cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE
        self.loss_weight = {"loss_box_reg": cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_WEIGHT}
        self.num_bbox_reg_classes = 1 if self.cls_agnostic_bbox_reg else self.num_classes


@ROI_HEADS_OUTPUT_REGISTRY.register()
class FastRCNNOutputLayers(_OutputLayersBase):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self._common(cfg, input_shape)
        self.cls_score = Linear(self.input_size, self.num_classes + 1)
        self.bbox_pred = Linear(self.input_size, self.num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        for l in [self.cls_score, self.bbox_pred]:
            nn.init.constant_(l.bias, 0)
        self._fused = _PackedCache()

    def forward(self, x):
        """-> (scores [M,K+1], deltas [M,4K|4]); both are column slices of one fused GEMM output."""
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        c, b = self.cls_score, self.bbox_pred
        def build():
            w = torch.cat([c.weight, b.weight], 0)
            bias = torch.cat([c.bias, b.bias], 0)
            padk = (-w.shape[0]) % 4  # float4-aligned rows for the vector epilogue
            if padk:
                w = torch.cat([w, w.new_zeros(padk, w.shape[1])], 0)
                bias = torch.cat([bias, bias.new_zeros(padk)], 0)
            return K.pack_linear(w, bias)

        y = K.linear(x.contiguous(), self._fused.get([c.weight, c.bias, b.weight, b.bias], build))
        k1 = self.num_classes + 1
        return y[:, :k1], y[:, k1:k1 + self.num_bbox_reg_classes * 4]


@ROI_HEADS_OUTPUT_REGISTRY.register()
class CosineSimOutputLayers(_OutputLayersBase):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self._common(cfg, input_shape)
        self.cls_score = Linear(self.input_size, self.num_classes + 1, bias=False)
        self.scale = cfg.MODEL.ROI_HEADS.COSINE_SCALE
        if self.scale == -1:
            self.scale = nn.Parameter(torch.ones(1) * 20.0)
        self.bbox_pred = Linear(self.input_size, self.num_bbox_reg_classes * 4)
        nn.init.normal_(self.cls_score.weight, std=0.01)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self._cos = _PackedCache()

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        x = x.contiguous()
        xn = K.rownorm(x, eps=1e-5, mode=0)
        # the reference renormalises cls_score.weight.data IN PLACE on every call (fast_rcnn.py:830-837)
        w = self.cls_score.weight
        K.rownorm(w.data, eps=1e-5, mode=0, out=w.data)
        scale = float(self.scale) if not isinstance(self.scale, nn.Parameter) else float(self.scale.item())

        def build():
            pc = K.pack_linear(w)
            pc.scale = torch.full((pc.K,), scale, device=w.device, dtype=torch.float32)
            return pc

        # rownorm wrote through `out=`, which does not bump torch's version counter: always rebuild
        self._cos.key = None
        scores = K.linear(xn, self._cos.get([w], build))
        deltas = self.bbox_pred(x)
        return scores, deltas


class _BoxOnlyBase(nn.Module):
    """Box-corrector output layer: one Linear(input, 4) (reference lvc/modeling/roi_heads/roi_heads_cascade.py:26-78
    BoxOnlyLayers, :80-211 BoxOnlyLayersCascade).  forward -> (num_classes, deltas [M,4])."""

    def __init__(self, cfg, input_shape, box2box_transform=None):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.bbox_pred = Linear(input_size, 4)
        nn.init.normal_(self.bbox_pred.weight, std=0.001)
        nn.init.constant_(self.bbox_pred.bias, 0)
        self.box2box_transform = box2box_transform or Box2BoxTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.lambda_ = cfg.MODEL.UBBR.LAMBDA
        self.iterate = cfg.MODEL.ROI_HEADS.NAME != "CascadeROIHeads"

    def forward(self, x):
        if x.dim() > 2:
            x = torch.flatten(x, start_dim=1)
        return self.num_classes, self.bbox_pred(x.contiguous())

    def losses(self, predictions, proposals):
        """reference roi_heads_cascade.py:141-163 (+ box_reg_loss :165-195) on lvc_giou_box_loss."""
        _, deltas = predictions
        gt_classes = torch.cat([p.gt_classes for p in proposals], 0)
        pboxes = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
        assert not pboxes.requires_grad, "Proposals should not require gradients"
        gboxes = torch.cat([(p.gt_boxes if p.has("gt_boxes") else p.proposal_boxes).tensor for p in proposals], 0)
        return {"loss_box_reg": _GIoUBoxLoss.apply(deltas, pboxes, gboxes, gt_classes, self)}

    def predict_boxes(self, predictions, proposals):
        """reference roi_heads_cascade.py:197-211: decoded boxes per image (no clipping here)."""
        _, deltas = predictions
        n = [len(p) for p in proposals]
        pboxes = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
        boxes = K.decode_boxes(deltas.detach().contiguous(), pboxes.view(1, -1, 4).contiguous(),
                               self.box2box_transform.weights, None)
        return boxes.view(-1, 4).split(n, dim=0)


@ROI_HEADS_OUTPUT_REGISTRY.register()
class BoxOnlyLayers(_BoxOnlyBase):
    pass


@ROI_HEADS_OUTPUT_REGISTRY.register()
class BoxOnlyLayersCascade(_BoxOnlyBase):
    pass


class _GIoUBoxLoss(torch.autograd.Function):
    """deltas -> scalar loss; the kernel emits d(loss)/d(deltas) in the forward pass."""

    @staticmethod
    def forward(ctx, deltas, pboxes, gboxes, gt_classes, layer):
        loss, dd = K.giou_box_loss(deltas, pboxes, gboxes, gt_classes, layer.num_classes,
                                   layer.box2box_transform.weights, layer.box2box_transform.scale_clamp,
                                   iterate=layer.iterate, lambda_=layer.lambda_)
        ctx.save_for_backward(dd)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dd,) = ctx.saved_tensors
        return dd * g, None, None, None, None


class _PredictorLoss(torch.autograd.Function):
    """FastRCNNOutputLayers forward + FastRCNNOutputs.losses (reference fast_rcnn.py:267-279, 296-359, 424-438) as one
    differentiable op: forward = fused cls|bbox GEMM + lvc_fast_rcnn_losses (which also emits dlogits/ddeltas);
    backward = dW = dY^T X on the same MFMA GEMM kernel, db = column sums, and -- when the box head trains --
    dX = [dlogits | ddeltas] [Wc ; Wb]."""

    @staticmethod
    def forward(ctx, x, wc, bc, wb, bb, proposals, gt_boxes, gt_classes, layer):
        scores, deltas = layer(x)
        out, dl, dd = K.fast_rcnn_losses(scores, deltas, proposals, gt_boxes, gt_classes, layer.num_classes,
                                         layer.box2box_transform.weights, layer.smooth_l1_beta)
        ctx.save_for_backward(x, dl, dd, wc, wb)
        ctx.has_bias = (bc is not None, bb is not None)
        pred = scores.argmax(dim=1)
        ctx.mark_non_differentiable(pred)
        return out[0], out[1], pred

    @staticmethod
    def backward(ctx, g_cls, g_box, _g_pred):
        x, dl, dd, wc, wb = ctx.saved_tensors
        R = x.shape[0]
        pad = (-R) % 32  # the GEMM contracts over R: multiple of the 32-wide k chunk
        xt = torch.zeros(x.shape[1], R + pad, device=x.device)
        xt[:, :R] = x.t()
        pc = K.pack_linear(xt, split=K.DGRAD_SPLIT)

        def dweight(dy, g):
            dyt = torch.zeros(dy.shape[1], R + pad, device=x.device)
            dyt[:, :R] = (dy * g).t()
            return K.linear(dyt, pc, split=K.DGRAD_SPLIT)  # [K_out, K_in]; gradients: range-free split

        dwc, dwb = dweight(dl, g_cls), dweight(dd, g_box)
        dbc = (dl * g_cls).sum(0) if ctx.has_bias[0] else None
        dbb = (dd * g_box).sum(0) if ctx.has_bias[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dx, _ = K.linear_backward(x, torch.cat([wc, wb], 0), torch.cat([dl * g_cls, dd * g_box], 1).contiguous(),
                                      need_dx=True, need_dw=False)
        return dx, dwc, dbc, dwb, dbb, None, None, None, None


class _CosinePredictorLoss(torch.autograd.Function):
    """CosineSimOutputLayers forward (reference fast_rcnn.py:811-841) + FastRCNNOutputs.losses as one differentiable op.
    The reference renormalises `cls_score.weight.data` in place before the product, so the classification weight's
    gradient is that of the plain product scale * xn W^T (no gradient through the weight normalisation); the input
    normalisation x / (|x| + 1e-5) is differentiated (`lvc_rownorm_backward`) when the box head trains."""

    @staticmethod
    def forward(ctx, x, wc, wb, bb, proposals, gt_boxes, gt_classes, layer):
        scores, deltas = layer(x)      # renormalises wc.data in place, as every reference forward does
        out, dl, dd = K.fast_rcnn_losses(scores, deltas, proposals, gt_boxes, gt_classes, layer.num_classes,
                                         layer.box2box_transform.weights, layer.smooth_l1_beta)
        ctx.save_for_backward(x, dl, dd, wc.detach().clone(), wb)
        ctx.scale = float(layer.scale)
        ctx.has_bias = bb is not None
        pred = scores.argmax(dim=1)
        ctx.mark_non_differentiable(pred)
        return out[0], out[1], pred

    @staticmethod
    def backward(ctx, g_cls, g_box, _g_pred):
        x, dl, dd, wc, wb = ctx.saved_tensors
        x = x.contiguous()
        xn = K.rownorm(x, eps=1e-5, mode=0)
        dls = (dl * (g_cls * ctx.scale)).contiguous()
        ddg = (dd * g_box).contiguous()
        need_x = ctx.needs_input_grad[0]
        dxn, dwc = K.linear_backward(xn, wc, dls, need_dx=need_x, need_dw=True)
        dx, dwb = K.linear_backward(x, wb, ddg, need_dx=need_x, need_dw=True)
        if need_x:
            dx = K.rownorm_backward(x, dxn, eps=1e-5, accumulate_into=dx.contiguous())
        dbb = K.colsum_rows(ddg) if ctx.has_bias else None
        return dx, dwc, dwb, dbb, None, None, None, None


def fast_rcnn_losses(layer, x, proposals, gt_boxes, gt_classes):
    """-> ({"loss_cls", "loss_box_reg"}, predicted classes) for a FastRCNNOutputLayers `layer`."""
    if layer.box_reg_loss_type != "smooth_l1":
        raise NotImplementedError("ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE '{}'".format(layer.box_reg_loss_type))
    lw = layer.loss_weight
    if isinstance(layer, CosineSimOutputLayers):
        if isinstance(layer.scale, nn.Parameter):
            raise NotImplementedError("training with a learnable COSINE_SCALE (-1) is not used by the shipped configs")
        lc, lb, pred = _CosinePredictorLoss.apply(x, layer.cls_score.weight, layer.bbox_pred.weight, layer.bbox_pred.bias,
                                                  proposals, gt_boxes, gt_classes, layer)
        return {"loss_cls": lc * lw.get("loss_cls", 1.0), "loss_box_reg": lb * lw.get("loss_box_reg", 1.0)}, pred
    if not isinstance(layer, FastRCNNOutputLayers):
        raise NotImplementedError("training of {} is not implemented".format(type(layer).__name__))
    lc, lb, pred = _PredictorLoss.apply(x, layer.cls_score.weight, layer.cls_score.bias, layer.bbox_pred.weight,
                                         layer.bbox_pred.bias, proposals, gt_boxes, gt_classes, layer)
    return {"loss_cls": lc * lw.get("loss_cls", 1.0), "loss_box_reg": lb * lw.get("loss_box_reg", 1.0)}, pred
