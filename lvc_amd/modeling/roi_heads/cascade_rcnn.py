"""CascadeROIHeads for the class-agnostic box corrector (reference lvc/modeling/roi_heads/cascade_rcnn.py:33-369):
3 stages of ROIAlign -> 3-FC head -> Linear(1024,4) -> decode (stage weights) -> clip.  Module names as in the
reference (`box_head.{k}.fc{1,2,3}`, `box_predictor.{k}.bbox_pred`).

Implemented: the inference branch used by `GeneralizedRCNNRegOnly` (`_forward_box_qe`, :167-203: correct a given set
of (box, class) pseudo-labels), batched on device: per stage one ROIAlign launch over all levels, three fused
bias+ReLU GEMMs, one 1024->4 GEMM and one decode+clip kernel; and the TRAINING branch (`_forward_box` :205-238,
`_match_and_label_boxes` :279-327, `_run_stage` :329-346, `_create_proposals_from_boxes` :348-369) with the GIoU
loss kernel and GEMM backward through the three heads, for a frozen trunk (`MODEL.BACKBONE.FREEZE`, the shipped
`cascade_ubbr_*_ft_*` yaml) or a training one (`cascade_ubbr_R_50_FPN_base.yaml`, FREEZE_AT 2: ROIAlign backward
into p2..p5, then the fused Conv2d autograd of layers/wrappers.py through FPN and res5..res3).  The evaluation branch of
the box-corrector training configs (`_forward_box` eval half with `reg_only`: subsamples with `randperm` even in eval,
SURVEY row 20) returns the corrected foreground boxes and the proposals they came from.
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


class _ScaleGradient(torch.autograd.Function):
    """reference cascade_rcnn.py:22-30: identity forward, gradient times `scale` backward."""

    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


@ROI_HEADS_REGISTRY.register()
class CascadeROIHeads(ROIHeads):
    relabel_ignored_gt = False   # detectron2's label_and_sample_proposals: no gt_ignores toggle

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
        """Reference signature (cascade_rcnn.py:143-165): training -> (proposals, losses); evaluation only through
        GeneralizedRCNNRegOnly."""
        if self.training:
            proposals = self.label_and_sample_proposals(proposals, targets, inference=False)
            return proposals, self._forward_box_train(features, proposals, targets)
        if global_cfg.get("MODEL", {}).get("META_ARCHITECTURE", None) == "GeneralizedRCNNRegOnly":
            return self._forward_box_qe(features, None, targets)
        if not self.reg_only:
            raise NotImplementedError("CascadeROIHeads evaluation is implemented for BoxOnlyLayersCascade only")
        # reference :147-149, :160-162: detectron2's label_and_sample_proposals (still `randperm`-subsamples in eval,
        # only the logging is skipped), then the reg_only branch of _forward_box
        proposals = self.label_and_sample_proposals(proposals, targets, inference=False, log=False)
        return self._forward_box_eval(features, proposals)

    def _forward_box_eval(self, features, proposals):
        """reference cascade_rcnn.py:205-227 + :248-266 (reg_only evaluation): run the cascade on the sampled
        proposals; the result per image holds the last stage's boxes (clipped) of the FOREGROUND rows, in row order, with
        score 1 and the matched class -- what fast_rcnn_inference(one-hot scores, thresh 0.1, nms 1.0, topk 1e10)
        followed by the argsort of the kept row indices produces.  Returns (results, filtered stage-0 proposals)."""
        feats = {f: to_nhwc(features[f]) for f in self.box_in_features}
        dev = feats[self.box_in_features[0]].device
        counts = [len(p) for p in proposals]
        B, R = len(proposals), max(max(counts), 1)
        boxes = torch.zeros(B, R, 4, device=dev)
        for i, p in enumerate(proposals):
            boxes[i, : counts[i]] = p.proposal_boxes.tensor
        sizes = torch.tensor([list(p.image_size) for p in proposals], dtype=torch.int32, device=dev)
        out = self.refine_boxes_batched(feats, boxes, sizes)
        results, kept = [], []
        for i, p in enumerate(proposals):
            fg = p.gt_classes < self.num_classes
            inst = Instances(p.image_size)
            inst.pred_boxes = Boxes(out[i, : counts[i]][fg])
            inst.scores = torch.ones(int(fg.sum()), device=dev)
            inst.pred_classes = p.gt_classes[fg]
            results.append(inst)
            kept.append(p[fg])
        return tuple(results), kept

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

    # ------------------------------------------------------------------ training (frozen trunk)
    batched_training = True      # class switch for A/B runs and tests (False: the per-image lists of `forward`)

    def can_batch_train(self, targets):
        return (self.batched_training and self.training and self.reg_only and not self.proposal_append_gt and self.batch_size_per_image <= 1024
                and targets is not None and len(targets) > 0 and all(0 < len(t) <= 512 for t in targets))

    def forward_train_batched(self, features, table, keep, gt, gt_off, targets, image_sizes):
        """The training branch of `forward` (label_and_sample_proposals + _forward_box_train) on the proposal generator's padded
        table, one device->host read (the logged sample counts) at the end instead of ~35 in the per-image lists: every stage works on
        the fixed [B, BATCH_SIZE_PER_IMAGE] row block; rows the reference would not have (fewer samples than the quota, boxes that
        come out of a stage empty) are labelled -1 -- the regression loss and its normaliser only see foreground rows
        (lvc_giou_box_loss), so the losses and every gradient equal the per-image path's.  table [B,P,4] / keep bool [B,P]: RBG.forward_batched."""
        from ...utils.events import get_event_storage
        from .fast_rcnn import _GIoUBoxLoss

        B, P = keep.shape
        dev = table.device
        K_, bs = self.num_classes, self.batch_size_per_image
        with torch.no_grad():
            gcls = torch.cat([t.gt_classes for t in targets]).to(torch.int64)
            base = gt_off[:-1].long()[:, None]
            gmax = max(gt.shape[0] - 1, 0)
            # stage 0: Matcher (IOU_THRESHOLDS) + subsample_labels over the kept rows
            m, lab = K.match_boxes_batched(gt, gt_off, B, table, None, self.proposal_matcher.user_thresholds, self.proposal_matcher.labels,
                                           self.proposal_matcher.allow_low_quality_matches)
            lab = torch.where(keep, lab, torch.full_like(lab, -1))
            key, seed = K.sampling_keys(B, P, dev)
            sel, cnt = K.subsample_batched(lab, key, int(bs * self.positive_sample_fraction), bs, seed=seed)
            boxes, _lg, cls, sm = K.roi_gather_sampled(table, torch.zeros(B, P, device=dev), m, sel, cnt, gcls, gt_off, K_)
            gtb = gt[(base + sm).clamp(max=gmax).view(-1)].view(B, bs, 4)
            sizes = torch.tensor([[s[0], s[1]] for s in image_sizes], dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
            stats = [cnt[:, 0].long(), cnt[:, 1].long()]
        feats = [to_nhwc(features[f]) for f in self.box_in_features]
        losses = {}
        for k in range(self.num_cascade_stages):
            pooled = self.box_pooler.pool_nhwc(feats, boxes)
            if pooled.requires_grad:   # reference :338: gradients of the stage are averaged into the trunk
                pooled = _ScaleGradient.apply(pooled, 1.0 / self.num_cascade_stages)
            h = self.box_head[k].forward_nhwc(pooled)
            pred = self.box_predictor[k]
            _, deltas = pred(h)
            losses["loss_box_reg_stage{}".format(k)] = _GIoUBoxLoss.apply(deltas, boxes.view(-1, 4), gtb.view(-1, 4), cls.view(-1).contiguous(), pred)
            if k + 1 == self.num_cascade_stages:
                break
            with torch.no_grad():
                # _create_proposals_from_boxes (decode, clip, drop empty) + _match_and_label_boxes of the next stage
                nb = K.decode_boxes(deltas.detach(), boxes, pred.box2box_transform.weights, sizes)
                alive = (cls >= 0) & ((nb[..., 2] - nb[..., 0]) > 0) & ((nb[..., 3] - nb[..., 1]) > 0)
                mt = self.proposal_matchers[k + 1]
                m, lab = K.match_boxes_batched(gt, gt_off, B, nb, None, mt.user_thresholds, mt.labels, mt.allow_low_quality_matches)
                gi = (base + m.long()).clamp(max=gmax)
                fg = alive & (lab == 1)
                cls = torch.where(fg, gcls[gi.view(-1)].view(B, bs), torch.where(alive, torch.full_like(cls, K_), torch.full_like(cls, -1)))
                gtb = gt[gi.view(-1)].view(B, bs, 4)
                boxes = torch.where(alive[..., None], nb, torch.zeros_like(nb))
                stats += [fg.sum(1), (alive & ~fg).sum(1)]
        st = torch.stack(stats, 0).sum(1).tolist()       # the one device->host read
        storage = get_event_storage()
        storage.put_scalar("roi_head/num_fg_samples", st[0] / B)
        storage.put_scalar("roi_head/num_bg_samples", st[1] / B)
        for k in range(1, self.num_cascade_stages):
            storage.put_scalar("stage{}/roi_head/num_fg_samples".format(k), st[2 * k] / B)
            storage.put_scalar("stage{}/roi_head/num_bg_samples".format(k), st[2 * k + 1] / B)
        return losses

    def _forward_box_train(self, features, proposals, targets):
        """reference cascade_rcnn.py:205-238 (training half)."""
        feats = [to_nhwc(features[f]) for f in self.box_in_features]
        image_sizes = [p.image_size for p in proposals]
        head_outputs = []
        prev = None
        for k in range(self.num_cascade_stages):
            if k > 0:
                proposals = self._create_proposals_from_boxes(prev, image_sizes)
                proposals = self._match_and_label_boxes(proposals, k, targets)
            predictions = self._run_stage(feats, proposals, k)
            prev = self.box_predictor[k].predict_boxes(predictions, proposals)
            head_outputs.append((self.box_predictor[k], predictions, proposals))
        losses = {}
        for stage, (predictor, predictions, props) in enumerate(head_outputs):
            for name, v in predictor.losses(predictions, props).items():
                losses[name + "_stage{}".format(stage)] = v
        return losses

    def _run_stage(self, feats_nhwc, proposals, stage):
        """reference :329-346.  `_ScaleGradient` (:22-30) scales the gradient flowing from the head back into the
        pooled features by 1/num_stages; with a frozen trunk that gradient is never formed and the pooled rows enter
        the head as constants."""
        counts = [len(p) for p in proposals]
        B, R = len(proposals), max(max(counts), 1)
        dev = feats_nhwc[0].device
        boxes = torch.zeros(B, R, 4, device=dev)
        for i, p in enumerate(proposals):
            boxes[i, : counts[i]] = p.proposal_boxes.tensor
        pooled = self.box_pooler.pool_nhwc(feats_nhwc, boxes)
        keep = torch.cat([torch.arange(c, device=dev) + i * R for i, c in enumerate(counts)])
        pooled = pooled[keep].contiguous()
        if pooled.requires_grad:   # reference :338: gradients of the stage are averaged into the trunk
            pooled = _ScaleGradient.apply(pooled, 1.0 / self.num_cascade_stages)
        h = self.box_head[stage].forward_nhwc(pooled)
        return self.box_predictor[stage](h)

    @torch.no_grad()
    def _match_and_label_boxes(self, proposals, stage, targets):
        """reference :279-327: label the stage's input boxes with this stage's IoU threshold (no subsampling)."""
        from ...utils.events import get_event_storage

        num_fg, num_bg = [], []
        for prop, tgt in zip(proposals, targets):
            gt = tgt.gt_boxes.tensor
            matched_idxs, labels = self.proposal_matchers[stage].match(gt, prop.proposal_boxes.tensor)
            if len(tgt) > 0:
                gt_classes = tgt.gt_classes[matched_idxs]
                gt_classes[labels == 0] = self.num_classes
                gt_boxes = Boxes(gt[matched_idxs])
            else:
                gt_classes = torch.zeros_like(matched_idxs) + self.num_classes
                gt_boxes = Boxes(gt.new_zeros((len(prop), 4)))
            prop.gt_classes = gt_classes
            prop.gt_boxes = gt_boxes
            num_fg.append(int((labels == 1).sum()))
            num_bg.append(labels.numel() - num_fg[-1])
        storage = get_event_storage()
        storage.put_scalar("stage{}/roi_head/num_fg_samples".format(stage), sum(num_fg) / len(num_fg))
        storage.put_scalar("stage{}/roi_head/num_bg_samples".format(stage), sum(num_bg) / len(num_bg))
        return proposals

    def _create_proposals_from_boxes(self, boxes, image_sizes):
        """reference :348-369: detach, clip, drop empty boxes (training only)."""
        out = []
        for b, size in zip(boxes, image_sizes):
            bx = Boxes(b.detach().clone())
            bx.clip(size)
            if self.training:
                bx = bx[bx.nonempty()]
            prop = Instances(size)
            prop.proposal_boxes = bx
            out.append(prop)
        return out
