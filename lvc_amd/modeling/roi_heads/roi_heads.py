"""ROI heads: `ROIHeads` base + `StandardROIHeads` (reference lvc/modeling/roi_heads/roi_heads.py:
28, 90-306, 483-629).  Sub-module names `box_pooler`, `box_head`, `box_predictor` as in the reference.

Inference launch plan for B images x R proposals (all on device, fixed shapes, no host sync):
  assign levels + build rois (1 launch) -> ROIAlign over all levels (1 launch, output [B*R,7,7,C])
  -> fc1/fc2 MFMA GEMMs with fused bias+ReLU -> fused cls|bbox GEMM -> lvc_fast_rcnn_inference.
"""
import torch
from torch import nn

from ... import kernels as K
from ...layers import ShapeSpec
from ...layers.layout import require_device, to_nhwc
from ...structures import Boxes, Instances
from ...utils.registry import Registry
from ..box_regression import Box2BoxTransform
from ..poolers import ROIPooler
from .box_head import build_box_head
from .fast_rcnn import ROI_HEADS_OUTPUT_REGISTRY

ROI_HEADS_REGISTRY = Registry("ROI_HEADS")


def build_roi_heads(cfg, input_shape):
    return ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, input_shape)


class ROIHeads(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        RH, BH = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        self.batch_size_per_image = RH.BATCH_SIZE_PER_IMAGE
        self.positive_sample_fraction = RH.POSITIVE_FRACTION
        self.test_score_thresh = RH.SCORE_THRESH_TEST
        self.test_nms_thresh = RH.NMS_THRESH_TEST
        self.test_detections_per_img = cfg.TEST.DETECTIONS_PER_IMAGE
        self.in_features = RH.IN_FEATURES
        self.num_classes = RH.NUM_CLASSES
        self.proposal_append_gt = RH.PROPOSAL_APPEND_GT
        self.feature_strides = {k: v.stride for k, v in input_shape.items()}
        self.feature_channels = {k: v.channels for k, v in input_shape.items()}
        self.cls_agnostic_bbox_reg = BH.CLS_AGNOSTIC_BBOX_REG
        self.smooth_l1_beta = BH.SMOOTH_L1_BETA
        self.box_reg_loss_type = BH.BBOX_REG_LOSS_TYPE
        self.loss_weight = {"loss_box_reg": BH.BBOX_REG_LOSS_WEIGHT}
        self.ignore_reg = RH.IGNORE_REG
        self.iou_thresholds, self.iou_labels = RH.IOU_THRESHOLDS, RH.IOU_LABELS
        self.box2box_transform = Box2BoxTransform(weights=BH.BBOX_REG_WEIGHTS)


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(ROIHeads):
    def __init__(self, cfg, input_shape):
        super().__init__(cfg, input_shape)
        self._init_box_head(cfg)
        if cfg.DEBUG:
            raise NotImplementedError("cfg.DEBUG heads (FastRCNNOutputsDebug) are not part of the hot path")
        if cfg.MODEL.META_ARCHITECTURE == "GeneralizedRCNN_Context":
            raise NotImplementedError("GeneralizedRCNN_Context is unused by every shipped config")
        self.rbg = cfg.MODEL.PROPOSAL_GENERATOR.NAME == "RBG"
        self.reg_off = cfg.MODEL.ROI_HEADS.REG_OFF

    def _init_box_head(self, cfg):
        BH = cfg.MODEL.ROI_BOX_HEAD
        res = BH.POOLER_RESOLUTION
        scales = tuple(1.0 / self.feature_strides[k] for k in self.in_features)
        in_channels = [self.feature_channels[f] for f in self.in_features]
        assert len(set(in_channels)) == 1, in_channels
        self.box_pooler = ROIPooler(output_size=res, scales=scales, sampling_ratio=BH.POOLER_SAMPLING_RATIO,
                                    pooler_type=BH.POOLER_TYPE)
        self.box_head = build_box_head(cfg, ShapeSpec(channels=in_channels[0], height=res, width=res))
        self.box_predictor = ROI_HEADS_OUTPUT_REGISTRY.get(cfg.MODEL.ROI_HEADS.OUTPUT_LAYER)(cfg, self.box_head.output_size)

    def forward_batched(self, feats_nhwc, prop_boxes, prop_count, image_sizes_dev, post=None, status=None):
        """feats_nhwc: dict name -> [B,H,W,C]; prop_boxes [B,R,4]; prop_count [B] int32 or None.
        Returns (boxes [B,topk,4], scores, classes int32, rows int32, count [B] int32), all on device."""
        if self.reg_off:
            raise NotImplementedError("ROI_HEADS.REG_OFF is not used by the shipped configs")
        feats = [feats_nhwc[f] for f in self.in_features]
        pooled = self.box_pooler.pool_nhwc(feats, prop_boxes, status=status)
        h = self.box_head.forward_nhwc(pooled)
        scores, deltas = self.box_predictor(h)
        return K.fast_rcnn_inference(scores, deltas, prop_boxes, prop_count, image_sizes_dev, self.num_classes,
                                     self.box2box_transform.weights, self.test_score_thresh, self.test_nms_thresh,
                                     self.test_detections_per_img, post=post, status=status)

    def forward(self, images, features, proposals, targets=None):
        """Reference signature (roi_heads.py:554-572): -> (list[Instances], losses)."""
        if self.training or self.rbg:
            raise NotImplementedError("ROI-head training / RBG evaluation is not implemented in lvc_amd round 1")
        feats = {f: to_nhwc(features[f]) for f in self.in_features}
        dev = feats[self.in_features[0]].device
        require_device(feats[self.in_features[0]], "StandardROIHeads")
        counts = [len(p) for p in proposals]
        B, R = len(proposals), max(max(counts), 1)
        boxes = torch.zeros(B, R, 4, device=dev)
        for i, p in enumerate(proposals):
            boxes[i, : counts[i]] = p.proposal_boxes.tensor
        sizes = torch.tensor([list(p.image_size) for p in proposals], dtype=torch.int32, device=dev)
        status = K.new_status(dev)
        ob, osc, ocl, orow, cnt = self.forward_batched(feats, boxes, torch.tensor(counts, dtype=torch.int32, device=dev),
                                                       sizes, status=status)
        return instances_from_batched(ob, osc, ocl, cnt, [p.image_size for p in proposals], status), {}


def instances_from_batched(boxes, scores, classes, count, image_sizes, status=None):
    """One device->host read (counts + status), then per-image `Instances` of device tensors."""
    if status is not None:
        meta = torch.cat([count, status]).tolist()
        counts, st = meta[:-1], meta[-1]
        check_status(st)
    else:
        counts = count.tolist()
    out = []
    for i, size in enumerate(image_sizes):
        n = counts[i]
        inst = Instances(tuple(size))
        inst.pred_boxes = Boxes(boxes[i, :n])
        inst.scores = scores[i, :n]
        inst.pred_classes = classes[i, :n].to(torch.int64)
        out.append(inst)
    return out


def check_status(st):
    """Device status word written by the kernels (no sync on the hot path; read with the results)."""
    if st & 1:
        raise RuntimeError("ROIs in ROIAlign cannot have non-negative size!")  # reference ROIAlign_cpu.cpp:149-152
    if st & 2:
        raise RuntimeError("lvc_fast_rcnn_inference: more than 16384 (roi, class) candidates above "
                           "SCORE_THRESH_TEST in one image; raise the threshold")
