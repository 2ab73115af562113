"""GeneralizedRCNN / ProposalNetwork (reference lvc/modeling/meta_arch/rcnn.py:25-333, 413-488).

`forward(batched_inputs: list[dict]) -> list[{"instances": Instances}]` in eval mode, same input keys
(`image`, optional `height`/`width`/`proposals`) and the same freeze switches as the reference.

The inference path is one uninterrupted stream of HIP launches with fixed shapes:
  preprocess (normalise + pad + NHWC4)  ->  ResNet/FPN (fp32-accurate split-operand MFMA implicit GEMM, fused epilogues)
  ->  RPN head + on-device proposal selection  ->  ROIAlign over all levels  ->  box head GEMMs
  ->  softmax / decode / per-class NMS / top-k / detector_postprocess
with exactly ONE device->host read at the end (per-image detection counts + the status word), where
the reference path has >= 10 hidden syncs per image.
"""
import torch
from torch import nn

from ... import kernels as K
from ...structures import Boxes, ImageList, Instances
from ..backbone import build_backbone
from ..postprocessing import detector_postprocess
from ..proposal_generator import RPN, build_proposal_generator
from ..roi_heads import StandardROIHeads, build_roi_heads
from ..roi_heads.roi_heads import check_status, instances_from_batched, run_with_fallbacks
from .build import META_ARCH_REGISTRY


def _freeze(module):
    for p in module.parameters():
        p.requires_grad = False


class _RCNNBase(nn.Module):
    def _dev_const(self, values, dtype):
        """Small per-batch constants (image sizes, post-process scales) as cached device tensors: the same shapes
        recur every batch, the H2D copy happens once, and the forward stays capturable in a hipGraph."""
        cache = self.__dict__.setdefault("_const_cache", {})
        key = (dtype, tuple(tuple(v) for v in values))
        t = cache.get(key)
        if t is None or t.device.type != torch.device(self.device).type:
            if len(cache) > 256:
                cache.clear()
            t = torch.tensor([list(v) for v in values], dtype=dtype, device=self.device)
            cache[key] = t
        return t

    def _init_common(self, cfg):
        self.device = torch.device(cfg.MODEL.DEVICE)
        assert len(cfg.MODEL.PIXEL_MEAN) == len(cfg.MODEL.PIXEL_STD)
        self.pixel_mean = [float(v) for v in cfg.MODEL.PIXEL_MEAN]
        self.pixel_std = [float(v) for v in cfg.MODEL.PIXEL_STD]
        assert len(self.pixel_mean) == 3, "the stem kernel is built for 3-channel images"
        self._input_cfg = cfg

    def preprocess_image(self, batched_inputs):
        """Normalize, pad and batch (reference rcnn.py:324-333).  Storage is NHWC with 4 channel slots;
        `.tensor` is the NCHW-shaped [N,3,Hp,Wp] view of it."""
        if "image" not in batched_inputs[0] and "raw" in batched_inputs[0]:
            return self._preprocess_raw(batched_inputs)
        images = [x["image"].to(self.device, non_blocking=True) for x in batched_inputs]
        sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        Hp, Wp = ImageList.padded_size(sizes, self.backbone.size_divisibility)
        buf = torch.empty(len(images), Hp, Wp, 4, device=self.device, dtype=torch.float32)
        if len({im.dtype for im in images}) == 1 and images[0].dtype in (torch.float32, torch.uint8):
            K.preprocess_batch_into(images, buf, self.pixel_mean, self.pixel_std)     # the whole batch in one launch
        else:
            for i, im in enumerate(images):
                K.preprocess_into(im, buf[i], self.pixel_mean, self.pixel_std)
        return ImageList(buf.permute(0, 3, 1, 2)[:, :3], sizes)

    def _preprocess_raw(self, batched_inputs):
        """Inputs that still carry the decoded file: {"raw": uint8 [H,W,3] in INPUT.FORMAT channel order, ...}.  The
        test-time ResizeShortestEdge of the reference's DatasetMapper (data/dataset_mapper.py:148-158 ->
        detection_utils.py:563-595) runs on the device, Pillow-exact, and writes the normalised pixels straight into the
        batch slot: 1 byte per sample crosses PCIe and no float CHW copy is formed."""
        from ...data import ResizeShortestEdge

        aug = self.__dict__.get("_test_resize")
        if aug is None:
            aug = self.__dict__["_test_resize"] = ResizeShortestEdge.from_config(self._input_cfg, is_train=False)
        raws = [x["raw"].to(self.device, non_blocking=True) for x in batched_inputs]
        tfms = [aug.get_transform(r) for r in raws]
        sizes = [(t.new_h, t.new_w) if t is not None else (int(r.shape[0]), int(r.shape[1])) for t, r in zip(tfms, raws)]
        Hp, Wp = ImageList.padded_size(sizes, self.backbone.size_divisibility)
        buf = torch.empty(len(raws), Hp, Wp, 4, device=self.device, dtype=torch.float32)
        for i, (r, t) in enumerate(zip(raws, tfms)):
            if t is None:
                K.preprocess_into(r.permute(2, 0, 1), buf[i], self.pixel_mean, self.pixel_std)
            else:
                t.apply_image(r, out_slot=buf[i], mean=self.pixel_mean, std=self.pixel_std)
        return ImageList(buf.permute(0, 3, 1, 2)[:, :3], sizes)


@META_ARCH_REGISTRY.register()
class GeneralizedRCNN(_RCNNBase):
    def __init__(self, cfg):
        super().__init__()
        self._init_common(cfg)
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.roi_heads = build_roi_heads(cfg, self.backbone.output_shape())
        self.to(self.device)
        M = cfg.MODEL
        if M.BACKBONE.FREEZE:
            _freeze(self.backbone)
        if M.BACKBONE.FREEZE_BOTTOM_UP:
            _freeze(self.backbone.bottom_up)
        if M.PROPOSAL_GENERATOR.FREEZE and self.proposal_generator:
            _freeze(self.proposal_generator)
        if M.ROI_HEADS.FREEZE_FEAT:
            _freeze(self.roi_heads.box_head)
            if M.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG:
                for n, p in self.roi_heads.box_predictor.named_parameters():
                    if "bbox_pred" in n:
                        p.requires_grad = False
        if M.ROI_HEADS.FREEZE_BBOX_PRED:
            for n, p in self.roi_heads.box_predictor.named_parameters():
                if "bbox_pred" in n:
                    p.requires_grad = False
        if M.PROPOSAL_GENERATOR.UNFREEZE_FIN:
            for p in self.proposal_generator.rpn_head.objectness_logits.parameters():
                p.requires_grad = True
            for p in self.proposal_generator.rpn_head.anchor_deltas.parameters():
                p.requires_grad = True
        assert not M.IMAGES_ONLY
        self.roi_heads_name = M.ROI_HEADS.NAME
        self.output_layer = M.ROI_HEADS.OUTPUT_LAYER

    def forward(self, batched_inputs):
        if not self.training:
            return self.inference(batched_inputs)

        def once():
            self.__dict__["_range_checked"] = False
            self._prepack_trainable()
            losses = self._forward_train(batched_inputs)
            if not self.__dict__["_range_checked"]:    # (the deferred-read path folds the range words into its one device->host read)
                K.check_conv_error_word(self.device)   # fp16x2 range word of the forward kernels (the step syncs anyway)
            return losses

        return run_with_fallbacks(self, once)

    def _prepack_trainable(self):
        """The kernel-layout operands of every convolution whose weight trains, rebuilt after an optimizer step in a few grouped
        launches (forward operand, and the data-gradient operand where a gradient flows further down) instead of 2-3 per layer."""
        if not K.PREPACK or not torch.is_grad_enabled():
            return
        lst = self.__dict__.get("_trainable_convs")
        if lst is None or lst[0] != sum(1 for p in self.parameters() if p.requires_grad):
            from ...layers import Conv2d

            convs = [m for m in self.modules() if isinstance(m, Conv2d) and m.weight.requires_grad]
            lst = self.__dict__["_trainable_convs"] = (sum(1 for p in self.parameters() if p.requires_grad), convs)
        if lst[1]:
            from ...layers import Conv2d

            Conv2d.prepack(lst[1], holder=self.__dict__.setdefault("_prepack_plan", {}))

    def _forward_train(self, batched_inputs):
        # training forward (reference rcnn.py:127-175): losses of the RPN (logged; frozen) and of the box predictor
        images = self.preprocess_image(batched_inputs)
        if "instances" in batched_inputs[0]:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        elif "targets" in batched_inputs[0]:
            gt_instances = [x["targets"].to(self.device) for x in batched_inputs]
        else:
            gt_instances = None
        # frozen trunk (the fine-tune yamls): no graph; otherwise the fused Conv2d autograd records res3.. / FPN
        with torch.set_grad_enabled(torch.is_grad_enabled() and any(p.requires_grad for p in self.backbone.parameters())):
            features = self.backbone(images.tensor)
        from ..proposal_generator.rbg import RBG

        if isinstance(self.proposal_generator, RBG):  # reference rcnn.py:150-155: loaded proposals -> jittered GT boxes
            proposals = [x["proposals"].to(self.device) for x in batched_inputs]
            proposal_losses = {}
            from ..roi_heads.cascade_rcnn import CascadeROIHeads

            if (isinstance(self.roi_heads, CascadeROIHeads) and self.roi_heads.can_batch_train(gt_instances)
                    and self.proposal_generator.can_batch(proposals, gt_instances)):
                # the box corrector's training rows as padded batch tensors: no per-image lists, one read for the logged counts
                gt, gt_off, _n = K.cat_ground_truth(gt_instances)
                table, keep = self.proposal_generator.forward_batched(proposals, gt_instances, gt, gt_off)
                return self.roi_heads.forward_train_batched(features, table, keep, gt, gt_off, gt_instances,
                                                            [x.image_size for x in gt_instances])
            proposals, _ = self.proposal_generator(proposals, gt_instances)
        elif (self.deferred_reads and isinstance(self.proposal_generator, RPN) and isinstance(self.roi_heads, StandardROIHeads)
              and self.proposal_generator.can_batch_targets(gt_instances) and self.roi_heads.can_batch_train(gt_instances)):
            return self._forward_train_deferred(images, features, gt_instances)
        elif self.proposal_generator:
            proposals, proposal_losses = self.proposal_generator(images, features, gt_instances)
        else:
            assert "proposals" in batched_inputs[0]
            proposals = [x["proposals"].to(self.device) for x in batched_inputs]
            proposal_losses = {}
        _, detector_losses = self.roi_heads(images, features, proposals, gt_instances)
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    deferred_reads = True      # class switch for A/B runs and tests (False: RPN.forward / ROIHeads.forward with their own reads)

    def _forward_train_deferred(self, images, features, gt_instances):
        """RPN + StandardROIHeads training (reference rcnn.py:127-175) as ONE stream of launches with ONE device->host read at the end:
        the proposal generator's batch tensors go straight to the heads (no per-image Instances in between), every count the step logs
        -- proposals, sampled anchors, sampled rows, accuracy, the conv kernels' range words -- comes back together.  If an image fell
        short of its ROI quota (the heads ran padded) the detector losses are recomputed by the per-image path: same values as the
        reference's normalisation in every case."""
        from ...utils.events import get_event_storage

        rpn, heads = self.proposal_generator, self.roi_heads
        boxes, logits, count, proposal_losses, rpn_counts, gt, gt_off = rpn.forward_train_batched(images, features, gt_instances)
        detector_losses, cnt, stats = heads.forward_train_batched(features, boxes, logits, count, gt, gt_off, gt_instances)
        B, bs = boxes.shape[0], heads.batch_size_per_image
        meta = torch.cat([count.long(), rpn_counts.view(-1).long(), cnt.view(-1).long(), stats.long(), K.range_summary(self.device).long()]).tolist()
        pc, rc, cc, st, flagged = meta[:B], meta[B:3 * B], meta[3 * B:5 * B], meta[5 * B:5 * B + 4], meta[-1]
        if flagged:
            K.check_conv_error_word(self.device)      # re-routes the layers concerned and raises: `run_with_fallbacks` repeats the pass
        self.__dict__["_range_checked"] = True
        storage = get_event_storage()
        storage.put_scalar("rpn/num_pos_anchors", sum(rc[0::2]) / B)
        storage.put_scalar("rpn/num_neg_anchors", sum(rc[1::2]) / B)
        counts = [(cc[2 * i], cc[2 * i + 1]) for i in range(B)]
        if all(a + b == bs for a, b in counts):
            heads.log_train_scalars(counts, st, B * bs)
        else:
            proposals = []
            for i, size in enumerate(images.image_sizes):
                inst = Instances(size)
                inst.proposal_boxes = Boxes(boxes[i, : pc[i]])
                inst.objectness_logits = logits[i, : pc[i]]
                proposals.append(inst)
            _, detector_losses = heads(images, features, proposals, gt_instances)
            K.check_conv_error_word(self.device)      # this second heads pass raised its own range words: judged in THIS step
        losses = {}
        losses.update(detector_losses)
        losses.update(proposal_losses)
        return losses

    # ------------------------------------------------------------------ device-side fast path
    def inference_batched(self, batched_inputs, do_postprocess=True):
        """Whole forward with device-resident, fixed-shape outputs and no host sync:
        (boxes [B,topk,4], scores [B,topk], classes [B,topk] int32, count [B] int32, status [1] int32)."""
        assert isinstance(self.proposal_generator, RPN) and isinstance(self.roi_heads, StandardROIHeads)
        images = self.preprocess_image(batched_inputs)
        sizes = images.image_sizes
        dev = self.device
        sizes_dev = self._dev_const(sizes, torch.int32)
        N, _, Hp, Wp = images.tensor.shape
        x4 = images.tensor.as_strided((N, Hp, Wp, 4), (Hp * Wp * 4, Wp * 4, 4, 1), images.tensor.storage_offset())
        feats = self.backbone.forward_nhwc(x4)
        pboxes, _plogits, pcount = self.proposal_generator.predict_proposals_batched(feats, sizes_dev)
        post = None
        if do_postprocess:
            rows = []
            for inp, (h, w) in zip(batched_inputs, sizes):
                dh, dw = (int(inp["raw"].shape[0]), int(inp["raw"].shape[1])) if "image" not in inp else (h, w)
                oh, ow = inp.get("height", dh), inp.get("width", dw)   # DatasetMapper: height/width = the file's size
                rows.append([ow / w, oh / h, float(oh), float(ow)])
            post = self._dev_const(rows, torch.float32)
        status = K.new_status(dev)
        ob, osc, ocl, _orow, cnt = self.roi_heads.forward_batched(feats, pboxes, pcount, sizes_dev, post=post, status=status)
        return ob, osc, ocl, cnt, status

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        """Reference rcnn.py:177-322 (the GeneralizedRCNN + RPN + StandardROIHeads branch)."""
        assert not self.training
        if detected_instances is not None:
            raise NotImplementedError("forward_with_given_boxes (mask/keypoint heads) is not on the box-only path")
        if getattr(self, "output_layer", None) == "BoxOnlyLayersCascade":
            return self._inference_box_corrector(batched_inputs)
        if not (isinstance(self.proposal_generator, RPN) and isinstance(self.roi_heads, StandardROIHeads)):
            return self._inference_modular(batched_inputs, do_postprocess)
        return run_with_fallbacks(self, lambda: self._inference_fast(batched_inputs, do_postprocess))

    def _inference_fast(self, batched_inputs, do_postprocess):
        ob, osc, ocl, cnt, status = self.inference_batched(batched_inputs, do_postprocess)
        out_sizes = []
        for inp in batched_inputs:
            if "image" in inp:
                h, w = int(inp["image"].shape[-2]), int(inp["image"].shape[-1])
                dh, dw = h, w
            else:   # raw file pixels: the network saw the ResizeShortestEdge size, results default to the file's size
                dh, dw = int(inp["raw"].shape[0]), int(inp["raw"].shape[1])
                t = self._test_resize.get_transform(inp["raw"])
                h, w = (t.new_h, t.new_w) if t is not None else (dh, dw)
            out_sizes.append((inp.get("height", dh), inp.get("width", dw)) if do_postprocess else (h, w))
        insts = instances_from_batched(ob, osc, ocl, cnt, out_sizes, status)
        return [{"instances": r} for r in insts] if do_postprocess else insts

    def _inference_box_corrector(self, batched_inputs):
        """reference rcnn.py:201-230: evaluation of the box corrector = IoU with the matched GT before and after the
        cascade, for the (sub)sampled foreground proposals.  Not a hot path: the IoUs come from the match kernel's
        pairwise form on the host side of the tensors."""
        from ...structures import pairwise_iou

        images = self.preprocess_image(batched_inputs)
        with torch.no_grad():
            features = self.backbone(images.tensor)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        out_results, in_proposals = self.roi_heads(images, features, proposals, gt_instances)
        input_ious, output_ious, gt_classes = [], [], []
        for out, inp in zip(out_results, in_proposals):
            gt = inp.gt_boxes
            input_ious.append(torch.diag(pairwise_iou(inp.proposal_boxes, gt)))
            output_ious.append(torch.diag(pairwise_iou(out.pred_boxes, gt)))
            gt_classes.append(inp.gt_classes)
        return {"input_ious": torch.cat(input_ious), "output_ious": torch.cat(output_ious), "gt_classes": torch.cat(gt_classes)}

    def _inference_modular(self, batched_inputs, do_postprocess):
        from ..proposal_generator.rbg import RBG

        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        if isinstance(self.proposal_generator, RBG):
            # reference rcnn.py:277-299: the loaded proposals and the ground truth go through RBG (identity in eval,
            # rbg.py:49-50) and the heads label / subsample them (roi_heads.py:561-562); the results are returned as the
            # heads made them -- `do_postprocess` is forced off for every head but UBBRROIHeads (:286)
            if "instances" not in batched_inputs[0] or "proposals" not in batched_inputs[0]:
                raise KeyError("PROPOSAL_GENERATOR.NAME = 'RBG' evaluates loaded `proposals` against `instances` (reference "
                               "rcnn.py:278-284); both keys are required in every input dict")
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            proposals = [x["proposals"].to(self.device) for x in batched_inputs]
            proposals, _ = self.proposal_generator(proposals, gt_instances)
            results, _ = self.roi_heads(images, features, proposals, gt_instances)
            if getattr(self, "roi_heads_name", None) != "UBBRROIHeads":
                return results
            return [{"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))}
                    for r, inp, size in zip(results, batched_inputs, images.image_sizes)]
        if self.proposal_generator:
            proposals, _ = self.proposal_generator(images, features, None)
        else:
            assert "proposals" in batched_inputs[0]
            proposals = [x["proposals"].to(self.device) for x in batched_inputs]
        results, _ = self.roi_heads(images, features, proposals, None)
        if not do_postprocess:
            return results
        processed = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            processed.append({"instances": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return processed


@META_ARCH_REGISTRY.register()
class ProposalNetwork(_RCNNBase):
    """reference rcnn.py:413-488: backbone + RPN only; output key "proposals"."""

    def __init__(self, cfg):
        super().__init__()
        self._init_common(cfg)
        self.backbone = build_backbone(cfg)
        self.proposal_generator = build_proposal_generator(cfg, self.backbone.output_shape())
        self.to(self.device)

    def forward(self, batched_inputs, no_post=False):
        """reference rcnn.py:433-480: training -> the RPN losses; `no_post` -> (proposals, images) as they are."""
        images = self.preprocess_image(batched_inputs)
        with torch.set_grad_enabled(self.training and torch.is_grad_enabled()
                                    and any(p.requires_grad for p in self.backbone.parameters())):
            features = self.backbone(images.tensor)
        if "instances" in batched_inputs[0]:
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        elif "targets" in batched_inputs[0]:
            gt_instances = [x["targets"].to(self.device) for x in batched_inputs]
        else:
            gt_instances = None
        proposals, proposal_losses = self.proposal_generator(images, features, gt_instances)
        if no_post:
            return proposals, images
        if self.training:
            return proposal_losses
        processed = []
        for r, inp, size in zip(proposals, batched_inputs, images.image_sizes):
            processed.append({"proposals": detector_postprocess(r, inp.get("height", size[0]), inp.get("width", size[1]))})
        return processed


@META_ARCH_REGISTRY.register()
class GeneralizedRCNNRegOnly(GeneralizedRCNN):
    """Box-corrector inference over given (box, class) pseudo-labels (reference rcnn.py:336-410): every input dict
    carries `instances` with `gt_boxes` / `gt_classes`; the output is the same dict with `pred_boxes` (corrected,
    rescaled to height/width, empty boxes dropped) and `pred_classes` set on its instances, `image` removed."""

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True):
        assert not self.training
        if detected_instances is not None:
            raise NotImplementedError("forward_with_given_boxes is not on the box-only path")
        images = self.preprocess_image(batched_inputs)
        features = self.backbone(images.tensor)
        gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        results, _ = self.roi_heads(images, features, None, gt_instances)
        processed = []
        for res, inp, size in zip(results, batched_inputs, images.image_sizes):
            inst = inp["instances"].to(self.device)
            inst.set("pred_boxes", res.pred_boxes)
            inst.set("pred_classes", inst.gt_classes)
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            inp["instances"] = detector_postprocess(inst, h, w)
            del inp["image"]
            processed.append(inp)
        return processed
