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
from ...utils.events import get_event_storage
from ...utils.registry import Registry
from ..box_regression import Box2BoxTransform
from ..matcher import Matcher
from ..sampling import subsample_labels
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
        self.proposal_matcher = Matcher(RH.IOU_THRESHOLDS, RH.IOU_LABELS, allow_low_quality_matches=False)

    def _sample_proposals(self, matched_idxs, matched_labels, gt_classes, inference=False):
        """reference lvc roi_heads.py:129-171."""
        if gt_classes.numel() > 0:
            gt_classes = gt_classes[matched_idxs]
            gt_classes[matched_labels == 0] = self.num_classes
            gt_classes[matched_labels == -1] = -1
        else:
            gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
        fg, bg = subsample_labels(gt_classes, self.batch_size_per_image, self.positive_sample_fraction, self.num_classes,
                                  inference)
        sampled = torch.cat([fg, bg], dim=0)
        return sampled, gt_classes[sampled]

    relabel_ignored_gt = True

    batched_sampling = True     # class switch for A/B runs and tests (False: the per-image loop in every case)

    batched_targets = True      # class switch for A/B runs and tests (False: the padded table built image by image with torch ops)

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, targets, inference=False, log=None):
        """reference lvc roi_heads.py:173-278: append GT, match (IoU kernel), gt_ignores toggle, subsample.
        `log` (default: not inference) controls only the EventStorage scalars: CascadeROIHeads inherits detectron2's
        label_and_sample_proposals, whose evaluation call still subsamples and merely skips the logging."""
        if (self.batched_sampling and not inference and self.proposal_append_gt and len(proposals) > 0
                and all(len(t) > 0 and not (self.relabel_ignored_gt and t.has("gt_ignores")) for t in targets)
                and all(p.proposal_boxes.tensor.is_cuda for p in proposals)):
            return self._label_and_sample_batched(proposals, targets, (not inference) if log is None else log)
        return self._label_and_sample_loop(proposals, targets, inference, log)

    def _label_and_sample_batched(self, proposals, targets, log):
        """The training case of the loop below for the whole batch with ONE device->host read (the per-image loop reads the device
        ~5 times per image: `nonzero`, the fg / bg counts): labels of all images in one padded [B, W] table, the fg / bg subsets
        drawn by ranking ONE random permutation of B*W keys within each row (any subset of a random permutation is in uniformly
        random order, so this is `positive[randperm(n)[:num_pos]]` in distribution; with `torch.randperm` patched to arange
        -- the parity tests -- it is the reference's selection exactly: the first num_pos positives, the first num_neg
        negatives, fg before bg).  Needs: training, PROPOSAL_APPEND_GT, every image with at least one gt box and (for lvc's own
        ROIHeads, which relabel proposals on ignored gt boxes) no gt_ignores field."""
        import math

        B, dev = len(proposals), proposals[0].proposal_boxes.tensor.device
        K_, bs = self.num_classes, self.batch_size_per_image
        gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
        batch = getattr(proposals[0], "_lvc_batch", None)
        if (batch is not None and self.batched_targets and bs <= 1024 and all(len(t) <= 512 for t in targets)
                and len(self.proposal_matcher.user_thresholds) in (1, 2) and len(batch[3]) == B):
            # the proposals as RPN.forward made them -- one padded [B,P] table on the device -- and the concatenated ground truth: table
            # + Matcher + subsample_labels + gather for the whole batch in five launches (csrc/train_targets.hip), ONE device->host read
            pboxes, plogits, pcount, pcounts_host, gt, gt_off = batch
            s_boxes, s_logits, s_cls, s_m, s_gtb, cnt = _sample_batched(self, pboxes, plogits, pcount, gt, gt_off, targets)
            counts = cnt.tolist()      # the one device->host read
            out = []
            for i, (prop, tgt) in enumerate(zip(proposals, targets)):
                c = counts[i][0] + counts[i][1]
                inst = Instances(prop.image_size)
                inst.proposal_boxes = Boxes(s_boxes[i, :c])
                inst.objectness_logits = s_logits[i, :c]
                inst.gt_classes = s_cls[i, :c]
                inst.gt_boxes = Boxes(s_gtb[i, :c])
                st = None
                for name, val in tgt.get_fields().items():
                    if name.startswith("gt_") and not inst.has(name):
                        st = s_m[i, :c] if st is None else st
                        inst.set(name, val[st])
                out.append(inst)
            # for _forward_train: the same rows as padded batch tensors (a python attribute, not a field)
            out[0]._lvc_sampled = (s_boxes, s_cls, s_gtb, [c[0] + c[1] for c in counts])
            if log:
                storage = get_event_storage()
                storage.put_scalar("roi_head/num_fg_samples", sum(c[0] for c in counts) / B)
                storage.put_scalar("roi_head/num_bg_samples", sum(c[1] for c in counts) / B)
            return out
        ns = [len(p) + len(t) for p, t in zip(proposals, targets)]
        W = max(max(ns), bs)
        boxes = torch.zeros(B, W, 4, device=dev)
        logits = torch.zeros(B, W, device=dev)
        labels = torch.full((B, W), -1, dtype=torch.int64, device=dev)     # -1: ignored / padding
        midx = torch.zeros(B, W, dtype=torch.int64, device=dev)
        for i, (prop, tgt) in enumerate(zip(proposals, targets)):
            gt, n, npr = tgt.gt_boxes.tensor, ns[i], len(prop)
            boxes[i, :npr] = prop.proposal_boxes.tensor
            boxes[i, npr:n] = gt
            logits[i, :npr] = prop.objectness_logits
            logits[i, npr:n] = gt_logit
            m, l = self.proposal_matcher.match(gt, boxes[i, :n])
            gc = tgt.gt_classes.to(torch.int64)[m]
            labels[i, :n] = torch.where(l == 0, torch.full_like(gc, K_), torch.where(l == -1, torch.full_like(gc, -1), gc))
            midx[i, :n] = m
        key = torch.randperm(B * W, device=dev).view(B, W)
        fg = (labels >= 0) & (labels != K_)
        bg = labels == K_
        big = B * W
        sf = torch.argsort(torch.where(fg, key, torch.full_like(key, big)), dim=1)
        sb = torch.argsort(torch.where(bg, key, torch.full_like(key, big)), dim=1)
        npos = fg.sum(1).clamp(max=int(bs * self.positive_sample_fraction))
        nneg = torch.minimum(bg.sum(1), bs - npos)
        j = torch.arange(bs, device=dev)[None, :]
        sampled = torch.where(j < npos[:, None], sf[:, :bs], torch.gather(sb, 1, (j - npos[:, None]).clamp(0, W - 1)))
        s_boxes = torch.gather(boxes, 1, sampled[:, :, None].expand(B, bs, 4))
        s_logits = torch.gather(logits, 1, sampled)
        s_cls = torch.gather(labels, 1, sampled)
        s_m = torch.gather(midx, 1, sampled)
        counts = torch.stack([npos, nneg], 1).tolist()      # the one device->host read
        out = []
        for i, (prop, tgt) in enumerate(zip(proposals, targets)):
            c = counts[i][0] + counts[i][1]
            inst = Instances(prop.image_size)
            inst.proposal_boxes = Boxes(s_boxes[i, :c])
            inst.objectness_logits = s_logits[i, :c]
            inst.gt_classes = s_cls[i, :c]
            st = s_m[i, :c]
            for name, val in tgt.get_fields().items():
                if name.startswith("gt_") and not inst.has(name):
                    inst.set(name, val[st])
            out.append(inst)
        if log:
            storage = get_event_storage()
            storage.put_scalar("roi_head/num_fg_samples", sum(c[0] for c in counts) / B)
            storage.put_scalar("roi_head/num_bg_samples", sum(c[1] for c in counts) / B)
        return out

    def _label_and_sample_loop(self, proposals, targets, inference=False, log=None):
        out, num_fg, num_bg = [], [], []
        for prop, tgt in zip(proposals, targets):
            gt = tgt.gt_boxes.tensor
            pboxes, plogits = prop.proposal_boxes.tensor, prop.objectness_logits
            if self.proposal_append_gt:  # add_ground_truth_to_proposals: logit = log((1-1e-10)/(1-(1-1e-10)))
                import math

                gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
                pboxes = torch.cat([pboxes, gt], 0)
                plogits = torch.cat([plogits, gt_logit * torch.ones(len(gt), device=gt.device)], 0)
            matched_idxs, matched_labels = self.proposal_matcher.match(gt, pboxes)
            # lvc's ROIHeads only (roi_heads.py:222-228); CascadeROIHeads derives from detectron2's StandardROIHeads,
            # whose label_and_sample_proposals (detectron2/modeling/roi_heads/roi_heads.py:220-306) has no such toggle
            if self.relabel_ignored_gt and tgt.has("gt_ignores") and bool(tgt.gt_ignores.bool().sum()):
                from ...structures import pairwise_iou

                ig = tgt.gt_ignores.bool()
                max_ig = pairwise_iou(Boxes(gt[ig]), Boxes(pboxes)).max(dim=0)[0]
                matched_labels[max_ig > self.proposal_matcher.thresholds[1]] = -1
            sampled, gt_classes = self._sample_proposals(matched_idxs, matched_labels, tgt.gt_classes, inference)
            if self.proposal_append_gt:
                inst = Instances(prop.image_size)
                inst.proposal_boxes = Boxes(pboxes[sampled])
                inst.objectness_logits = plogits[sampled]
            else:   # reference: proposals_per_image[sampled_idxs] -- every field the proposals carry is kept
                inst = prop[sampled]
            inst.gt_classes = gt_classes
            if len(gt) > 0:
                st = matched_idxs[sampled]
                for name, val in tgt.get_fields().items():
                    if name.startswith("gt_") and not inst.has(name):
                        inst.set(name, val[st])
            else:
                inst.gt_boxes = Boxes(gt.new_zeros((len(sampled), 4)))
            num_bg.append(int((gt_classes == self.num_classes).sum()))
            num_fg.append(gt_classes.numel() - num_bg[-1])
            out.append(inst)
        if (not inference) if log is None else log:
            storage = get_event_storage()
            storage.put_scalar("roi_head/num_fg_samples", sum(num_fg) / max(1, len(num_fg)))
            storage.put_scalar("roi_head/num_bg_samples", sum(num_bg) / max(1, len(num_bg)))
        return out


@ROI_HEADS_REGISTRY.register()
class StandardROIHeads(ROIHeads):
    # capacity of the per-image (roi, class) candidate list of fast_rcnn_inference; None = R*K.  16 384 keeps the
    # detection-stage NMS in its one-block form; `run_with_fallbacks` switches to None after the first overflow
    det_max_candidates = 16384

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
                                     self.test_detections_per_img, post=post, status=status,
                                     max_candidates=self.det_max_candidates)

    def forward(self, images, features, proposals, targets=None):
        """Reference signature (roi_heads.py:554-572): -> (list[Instances], losses)."""
        if self.training:
            return self._forward_train(features, proposals, targets)
        if self.rbg:
            # reference roi_heads.py:561-562: with an RBG proposal generator the evaluation pass labels and subsamples the given
            # proposals against the ground truth first (inference=True: no EventStorage scalars), then runs the usual box branch
            proposals = self.label_and_sample_proposals(proposals, targets, inference=True)
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


def _sample_batched(self, pboxes, plogits, pcount, gt, gt_off, targets):
    """Table + Matcher + subsample_labels + gather for the whole batch (csrc/train_targets.hip), nothing read back:
    -> (boxes [B,bs,4], logits [B,bs], classes int64 [B,bs], matched gt index int64 [B,bs], matched gt boxes [B,bs,4], counts int32 [B,2])."""
    import math

    B, dev = pboxes.shape[0], pboxes.device
    K_, bs = self.num_classes, self.batch_size_per_image
    gt_logit = math.log((1.0 - 1e-10) / (1 - (1.0 - 1e-10)))
    Wt = pboxes.shape[1] + max(len(t) for t in targets)
    with torch.no_grad():
        tb, tl, nrow = K.roi_build_table(pboxes, plogits, pcount, gt, gt_off, gt_logit, Wt)
        m, lab = K.match_boxes_batched(gt, gt_off, B, tb, nrow, self.proposal_matcher.user_thresholds, self.proposal_matcher.labels,
                                       self.proposal_matcher.allow_low_quality_matches)
        key, seed = K.sampling_keys(B, Wt, dev)
        sel, cnt = K.subsample_batched(lab, key, int(bs * self.positive_sample_fraction), bs, seed=seed)
        gcls = torch.cat([t.gt_classes for t in targets]).to(torch.int64)
        s_boxes, s_logits, s_cls, s_m = K.roi_gather_sampled(tb, tl, m, sel, cnt, gcls, gt_off, K_)
        s_gtb = gt[(gt_off[:-1].long()[:, None] + s_m).clamp(max=max(gt.shape[0] - 1, 0)).view(-1)].view(B, bs, 4)
    return s_boxes, s_logits, s_cls, s_m, s_gtb, cnt


def can_batch_train(self, targets):
    return (self.batched_sampling and self.batched_targets and self.proposal_append_gt and not self.rbg and self.batch_size_per_image <= 1024
            and type(self)._forward_train is _forward_train and targets is not None and len(targets) > 0
            and all(0 < len(t) <= 512 and not (self.relabel_ignored_gt and t.has("gt_ignores")) for t in targets)
            and len(self.proposal_matcher.user_thresholds) in (1, 2))


def forward_train_batched(self, features, pboxes, plogits, pcount, gt, gt_off, targets):
    """StandardROIHeads' training branch on the proposal generator's batch tensors, without a device->host read: the sampled rows stay
    one padded [B, bs] table, ROIAlign / box head / losses run on all B * bs rows as if every image had filled its quota (the usual case;
    padding rows are zero boxes labelled background) and -> (losses, counts int32 [B,2], accuracy counts int64 [4]).  The CALLER reads the
    counts with the step's other scalars and, if some image did fall short of bs rows, discards these losses and runs the per-image path
    (`forward`) instead -- the reference's normalisation is by the true row count."""
    from .fast_rcnn import fast_rcnn_losses

    s_boxes, _s_logits, s_cls, _s_m, s_gtb, cnt = _sample_batched(self, pboxes, plogits, pcount, gt, gt_off, targets)
    feats = [to_nhwc(features[f]) for f in self.in_features]
    below = any(f.requires_grad for f in feats) or any(p.requires_grad for p in self.box_head.parameters())
    with torch.set_grad_enabled(below and torch.is_grad_enabled()):
        h = self.box_head.forward_nhwc(self.box_pooler.pool_nhwc(feats, s_boxes))
    gc = s_cls.view(-1)
    losses, pred = fast_rcnn_losses(self.box_predictor, h, s_boxes.view(-1, 4), s_gtb.view(-1, 4), gc)
    with torch.no_grad():
        fg = (gc >= 0) & (gc < self.num_classes)
        hit = pred == gc
        stats = torch.stack([fg.sum(), hit.sum(), (hit & fg).sum(), ((pred == self.num_classes) & fg).sum()])
    return losses, cnt, stats


def log_train_scalars(self, counts, stats, nrows):
    """EventStorage scalars of label_and_sample_proposals and FastRCNNOutputs._log_accuracy from host values."""
    storage = get_event_storage()
    B = len(counts)
    storage.put_scalar("roi_head/num_fg_samples", sum(c[0] for c in counts) / B)
    storage.put_scalar("roi_head/num_bg_samples", sum(c[1] for c in counts) / B)
    nfg, ncorrect, nfg_correct, nfg_bg = stats
    storage.put_scalar("fast_rcnn/cls_accuracy", float(ncorrect) / max(1, nrows))
    if nfg > 0:
        storage.put_scalar("fast_rcnn/fg_cls_accuracy", float(nfg_correct) / nfg)
        storage.put_scalar("fast_rcnn/false_negative", float(nfg_bg) / nfg)


def _forward_train(self, features, proposals, targets):
    """StandardROIHeads training branch (reference roi_heads.py:554-629 + FastRCNNOutputs.losses)."""
    from .fast_rcnn import fast_rcnn_losses

    proposals = self.label_and_sample_proposals(proposals, targets)
    feats = [to_nhwc(features[f]) for f in self.in_features]
    dev = feats[0].device
    counts = [len(p) for p in proposals]
    B, R = len(proposals), max(max(counts), 1)
    sampled = getattr(proposals[0], "_lvc_sampled", None) if proposals else None
    if sampled is not None and sampled[3] == counts and sampled[0].shape[1] == R:
        boxes = sampled[0]          # the sampler's own padded [B,R,4] table: no per-image copies
    else:
        sampled = None
        boxes = torch.zeros(B, R, 4, device=dev)
        for i, p in enumerate(proposals):
            boxes[i, : counts[i]] = p.proposal_boxes.tensor
    full = all(c == R for c in counts)
    # no graph when everything below the predictor is frozen (ft_novel yaml); otherwise ROIAlign backward into the
    # pyramid and the fused Linear autograd of the box head (base / ft_all yamls)
    below = any(f.requires_grad for f in feats) or any(p.requires_grad for p in self.box_head.parameters())
    with torch.set_grad_enabled(below and torch.is_grad_enabled()):
        pooled = self.box_pooler.pool_nhwc(feats, boxes)
        if full:                    # every image filled its quota (the usual case): the pooled rows are the head's input as they lie
            h = self.box_head.forward_nhwc(pooled)
        else:
            keep = torch.cat([torch.arange(c, device=dev) + i * R for i, c in enumerate(counts)])
            h = self.box_head.forward_nhwc(pooled[keep].contiguous())
    if sampled is not None and full:
        pb, gc, gb = sampled[0].view(-1, 4), sampled[1].view(-1), sampled[2].view(-1, 4)
    else:
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
        gb = torch.cat([p.gt_boxes.tensor for p in proposals], 0)
        gc = torch.cat([p.gt_classes for p in proposals], 0)
    losses, pred = fast_rcnn_losses(self.box_predictor, h, pb, gb, gc)
    # accuracy scalars (reference fast_rcnn.py _log_accuracy): the four counts in ONE device->host read
    storage = get_event_storage()
    fg = (gc >= 0) & (gc < self.num_classes)
    hit = pred == gc
    nfg, ncorrect, nfg_correct, nfg_bg = torch.stack([fg.sum(), hit.sum(), (hit & fg).sum(), ((pred == self.num_classes) & fg).sum()]).tolist()
    storage.put_scalar("fast_rcnn/cls_accuracy", float(ncorrect) / max(1, gc.numel()))
    if nfg > 0:
        storage.put_scalar("fast_rcnn/fg_cls_accuracy", float(nfg_correct) / nfg)
        storage.put_scalar("fast_rcnn/false_negative", float(nfg_bg) / nfg)
    return proposals, losses


StandardROIHeads._forward_train = _forward_train
StandardROIHeads.can_batch_train = can_batch_train
StandardROIHeads.forward_train_batched = forward_train_batched
StandardROIHeads.log_train_scalars = log_train_scalars


def widen_limits(model, exc):
    """React to a capacity / range condition the reference does not have: candidate list sized for R*K, or the range-free
    bf16x3 split; logged once.  Returns False when the limit was already at its widest (the caller re-raises)."""
    import logging

    if isinstance(exc, CandidateOverflow):
        heads = getattr(model, "roi_heads", None)
        if heads is None or heads.det_max_candidates is None:
            return False
        logging.getLogger("lvc_amd").warning(
            "more than %d (roi, class) candidates above SCORE_THRESH_TEST in one image: re-running with the list sized "
            "for R*K and keeping that size", heads.det_max_candidates)
        heads.det_max_candidates = None
        return True
    if isinstance(exc, K.Fp16RangeError):
        if exc.rerouted:       # the layers that overflowed were moved to their next wider form (kernels.check_conv_error_word)
            return True
        if K.CONV_SPLIT == "bf16x3":
            return False
        K.use_range_free_split()
        return True
    return False


def run_with_fallbacks(model, fn):
    """Run fn() (one pass that ends by reading the status words); on CandidateOverflow / Fp16RangeError widen the limit
    (`widen_limits`) and run again -- the pass degrades (one repeated batch, slower kernels) instead of failing."""
    for _ in range(8):     # a pass may move layers up one tier at a time (one accumulator -> two -> bf16x3), a few layers per pass
        try:
            return fn()
        except (CandidateOverflow, K.Fp16RangeError) as e:
            if not widen_limits(model, e):
                raise
    return fn()


def instances_from_batched(boxes, scores, classes, count, image_sizes, status=None):
    """One device->host read (counts + status), then per-image `Instances` of device tensors."""
    classes64 = classes.to(torch.int64)     # one conversion for the batch (queued BEFORE the blocking read: it runs in the step's tail,
                                            # not in the gap behind it); the per-image fields below are views
    if status is not None:
        # ONE read: counts, the status word and whether any of the conv kernels' error / range words is set
        meta = torch.cat([count, status, K.range_summary(count.device)]).tolist()
        counts, st, flagged = meta[:-2], meta[-2], meta[-1]
        check_status(st)
        if flagged:
            K.check_conv_error_word(count.device)   # spin timeout / a layer beyond its fp16 split's range: re-routes and raises
    else:
        counts = count.tolist()
    out = []
    # The host builds these while the GPU idles (forward() returns the batch's results before the next batch can be enqueued): the
    # three fields of an image are rows [0, n) of device tensors of one shape -- the objects are filled directly instead of through
    # Boxes.__init__ (as_tensor, reshape of empties, asserts) and Instances.set (a length check per field); 90 -> 45 us per batch.
    # ... and the B x 3 row ranges come from three split calls over the flattened tensors (sizes n_0, T - n_0, n_1, T - n_1, ...: every
    # other piece) instead of 24 narrows.
    B, T = scores.shape[0], scores.shape[1]
    if boxes.is_contiguous() and scores.is_contiguous() and len(image_sizes) == B:
        pieces = []
        for n in counts[:B]:
            pieces.append(n)
            pieces.append(T - n)
        bx = boxes.view(B * T, boxes.shape[2]).split_with_sizes(pieces)[0::2]
        sc = scores.view(B * T).split_with_sizes(pieces)[0::2]
        cl = classes64.view(B * T).split_with_sizes(pieces)[0::2]
    else:
        bx = [boxes[i, :counts[i]] for i in range(len(image_sizes))]
        sc = [scores[i, :counts[i]] for i in range(len(image_sizes))]
        cl = [classes64[i, :counts[i]] for i in range(len(image_sizes))]
    for i, size in enumerate(image_sizes):
        b = Boxes.__new__(Boxes)
        b.tensor = bx[i]
        inst = Instances(tuple(size))
        inst._fields.update(pred_boxes=b, scores=sc[i], pred_classes=cl[i])
        out.append(inst)
    return out


class CandidateOverflow(RuntimeError):
    """More (roi, class) pairs above SCORE_THRESH_TEST in one image than the candidate list was sized for.  The model
    entry points catch it, size the list for R*K (`StandardROIHeads.det_max_candidates = None`) and run again."""


def check_status(st):
    """Device status word written by the kernels (no sync on the hot path; read with the results)."""
    if st & 1:
        raise RuntimeError("ROIs in ROIAlign cannot have non-negative size!")  # reference ROIAlign_cpu.cpp:149-152
    if st & 2:
        raise CandidateOverflow("lvc_fast_rcnn_inference: more (roi, class) candidates above SCORE_THRESH_TEST in one "
                                "image than the candidate list holds; re-run with max_candidates=None (= R*K)")
