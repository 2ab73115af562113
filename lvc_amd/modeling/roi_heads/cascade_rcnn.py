"""CascadeROIHeads for the class-agnostic box corrector (reference lvc/modeling/roi_heads/cascade_rcnn.py:33-369):
3 stages of ROIAlign -> 3-FC head -> Linear(1024,4) -> decode (stage weights) -> clip.  Module names as in the
reference (`box_head.{k}.fc{1,2,3}`, `box_predictor.{k}.bbox_pred`).

Implemented: the inference branch used by `GeneralizedRCNNRegOnly` (`_forward_box_qe`, :167-203: correct a given set
of (box, class) pseudo-labels), batched on device: per stage one ROIAlign launch over all levels, three fused
bias+ReLU GEMMs, one 1024->4 GEMM and one decode+clip kernel.  Training (GIoU losses, `_ScaleGradient`, backward
into the trunk) and the RBG evaluation branch are not implemented.
"""
import torch
from torch import nn

from ... import kernels as K
from ...config import global_cfg
from ...layers import ShapeSpec
from ...layers.layout import to_nhwc
from ...structures import Boxes, Instances
from ..box_regression import Box2BoxTransform
from ..matcher import Matcher
from ..poolers import ROIPooler
from .box_head import build_box_head
from .fast_rcnn import ROI_HEADS_OUTPUT_REGISTRY
from .roi_heads import ROI_HEADS_REGISTRY, ROIHeads


@ROI_HEADS_REGISTRY.register()
class CascadeROIHeads(ROIHeads):
    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        BH = cfg.MODEL.ROI_BOX_HEAD
        res = BH.POOLER_RESOLUTION
        self.box_in_features = self.in_features
        scales = tuple(1.0 / input_shape[k].stride for k in self.in_features)
        weights = cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS
        ious = cfg.MODEL.ROI_BOX_CASCADE_HEAD.IOUS
        assert len(weights) == len(ious)
        assert BH.CLS_AGNOSTIC_BBOX_REG, "CascadeROIHeads only support class-agnostic regression now!"
        assert ious[0] == cfg.MODEL.ROI_HEADS.IOU_THRESHOLDS[0]
        output_layer = cfg.MODEL.ROI_HEADS.OUTPUT_LAYER
        if output_layer not in ("BoxOnlyLayersCascade", "BoxOnlyLayers"):
            raise NotImplementedError("CascadeROIHeads with classification heads is not used by the shipped configs")
        in_channels = [input_shape[f].channels for f in self.in_features]
        assert len(set(in_channels)) == 1, in_channels
        self.box_pooler = ROIPooler(output_size=res, scales=scales, sampling_ratio=BH.POOLER_SAMPLING_RATIO,
                                    pooler_type=BH.POOLER_TYPE)
        pooled = ShapeSpec(channels=in_channels[0], width=res, height=res)
        heads, preds, matchers = [], [], []
        for iou, w in zip(ious, weights):
            head = build_box_head(cfg, pooled)
            heads.append(head)
            preds.append(ROI_HEADS_OUTPUT_REGISTRY.get(output_layer)(cfg, head.output_size,
                                                                     box2box_transform=Box2BoxTransform(weights=w)))
            matchers.append(Matcher([iou], [0, 1], allow_low_quality_matches=False))
        self.num_cascade_stages = len(heads)
        self.box_head = nn.ModuleList(heads)
        self.box_predictor = nn.ModuleList(preds)
        self.proposal_matchers = matchers
        self.reg_only = output_layer == "BoxOnlyLayersCascade"

    def refine_boxes_batched(self, feats_nhwc, boxes, image_sizes_dev):
        """boxes [B,R,4] (zero rows = padding) -> corrected boxes [B,R,4] after all cascade stages."""
        feats = [feats_nhwc[f] for f in self.box_in_features]
        cur = boxes.contiguous()
        for k in range(self.num_cascade_stages):
            pooled = self.box_pooler.pool_nhwc(feats, cur)
            h = self.box_head[k].forward_nhwc(pooled)
            _, deltas = self.box_predictor[k](h)
            cur = K.decode_boxes(deltas, cur, self.box_predictor[k].box2box_transform.weights, image_sizes_dev)
        return cur

    def forward(self, images, features, proposals, targets=None):
        """Reference signature (cascade_rcnn.py:143-165).  Only the GeneralizedRCNNRegOnly evaluation branch."""
        if self.training:
            raise NotImplementedError("CascadeROIHeads training is not implemented in lvc_amd")
        if global_cfg.get("MODEL", {}).get("META_ARCHITECTURE", None) != "GeneralizedRCNNRegOnly":
            raise NotImplementedError("CascadeROIHeads evaluation with RBG proposals is not implemented; use "
                                      "META_ARCHITECTURE GeneralizedRCNNRegOnly (and call set_global_cfg(cfg))")
        return self._forward_box_qe(features, None, targets)

    def _forward_box_qe(self, features, proposals, targets):
        """reference cascade_rcnn.py:167-203: proposals = the targets' gt_boxes; output = corrected boxes of every
        target whose class is a foreground class, in input order, score 1, class = the given class."""
        feats = {f: to_nhwc(features[f]) for f in self.box_in_features}
        dev = feats[self.box_in_features[0]].device
        counts = [len(t) for t in targets]
        B, R = len(targets), max(max(counts), 1)
        boxes = torch.zeros(B, R, 4, device=dev)
        for i, t in enumerate(targets):
            t.set("proposal_boxes", t.gt_boxes)
            boxes[i, : counts[i]] = t.gt_boxes.tensor
        sizes = torch.tensor([list(t.image_size) for t in targets], dtype=torch.int32, device=dev)
        out = self.refine_boxes_batched(feats, boxes, sizes)
        results = []
        for i, t in enumerate(targets):
            keep = t.gt_classes < self.num_classes
            inst = Instances(t.image_size)
            inst.pred_boxes = Boxes(out[i, : counts[i]][keep])
            inst.scores = torch.ones(int(keep.sum()), device=dev)
            inst.pred_classes = t.gt_classes[keep]
            results.append(inst)
        return tuple(results), None
