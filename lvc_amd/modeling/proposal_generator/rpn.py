"""RPN: `StandardRPNHead` + `RPN` (reference detectron2/modeling/proposal_generator/rpn.py:67-139,
142-508; `find_top_rpn_proposals` proposal_utils.py:13-118).

Launch plan (inference), all on the current stream, no host sync:
  per level: 3x3 conv + bias + ReLU (MFMA implicit GEMM), then ONE fused 1x1 conv producing
  [B,H,W,A+4A] = objectness | deltas (the reference runs two convs);
  then lvc_rpn_proposals: per-(image,level) radix-select top-k, decode only the selected anchors,
  clip, drop empty, batched NMS (ballot kernels), gather top post_nms_topk.
"""
import torch
import torch.nn.functional as F
from torch import nn

from ... import kernels as K
from ...layers import Conv2d, ShapeSpec
from ...layers.layout import require_device, to_nchw_view, to_nhwc
from ...layers.wrappers import _PackedCache
from ...structures import Boxes, Instances
from ...utils.registry import Registry
from ...utils.events import get_event_storage
from ..anchor_generator import build_anchor_generator
from ..box_regression import Box2BoxTransform
from ..matcher import Matcher
from ..sampling import subsample_labels
from .build import PROPOSAL_GENERATOR_REGISTRY

RPN_HEAD_REGISTRY = Registry("RPN_HEAD")


def build_rpn_head(cfg, input_shape):
    return RPN_HEAD_REGISTRY.get(cfg.MODEL.RPN.HEAD_NAME)(cfg, input_shape)


MERGE_LEVELS = True      # the predictor of all pyramid levels as one launch (StandardRPNHead.forward_nhwc)
MERGE_LEVELS_CONV = True     # ... and the head's 3x3 conv over the levels as one launch
FUSE_PREDICTOR = True        # ... with the predictor in that launch's epilogue: the hidden maps are never written (kernels.conv3x3_levels_pred)


@RPN_HEAD_REGISTRY.register()
class StandardRPNHead(nn.Module):
    def __init__(self, cfg=None, input_shape=None, *, in_channels=None, num_anchors=None, box_dim=4):
        super().__init__()
        if cfg is not None:
            in_channels = [s.channels for s in input_shape]
            assert len(set(in_channels)) == 1, "Each level must have the same channel!"
            in_channels = in_channels[0]
            ag = build_anchor_generator(cfg, input_shape)
            num_anchors, box_dim = ag.num_anchors, ag.box_dim
            assert len(set(num_anchors)) == 1, "Each level must have the same number of anchors per spatial position"
            num_anchors = num_anchors[0]
        self.num_anchors, self.box_dim = num_anchors, box_dim
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1, activation=F.relu_)
        self.conv.two_acc = True   # logits / deltas decide top-k and NMS: keep the two-accumulator 3x3 form (kernels.HALO_S1)
        self.objectness_logits = Conv2d(in_channels, num_anchors, kernel_size=1, stride=1)
        self.anchor_deltas = Conv2d(in_channels, num_anchors * box_dim, kernel_size=1, stride=1)
        for l in [self.conv, self.objectness_logits, self.anchor_deltas]:
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self._fused = _PackedCache()
        self._fused_dgrad = _PackedCache()

    def fused_predictor(self):
        o, d = self.objectness_logits, self.anchor_deltas
        def build():
            w = torch.cat([o.weight, d.weight], 0)
            b = torch.cat([o.bias, d.bias], 0)
            padk = (-w.shape[0]) % 4  # zero output channels so rows are float4-aligned (vector epilogue)
            if padk:
                w = torch.cat([w, w.new_zeros((padk,) + tuple(w.shape[1:]))], 0)
                b = torch.cat([b, b.new_zeros(padk)], 0)
            return K.pack_conv(w, bias=b)

        return self._fused.get([o.weight, o.bias, d.weight, d.bias], build)

    def forward_nhwc(self, feats):
        """feats: list of [B,H,W,C] -> list of fused [B,H,W,A+A*box_dim] tensors."""
        pc = self.fused_predictor()
        o, d = self.objectness_logits, self.anchor_deltas
        train = torch.is_grad_enabled() and any(p.requires_grad for p in (o.weight, o.bias, d.weight, d.bias))
        grad_free = not train and not (torch.is_grad_enabled() and (any(x.requires_grad for x in feats)
                                                                    or any(p.requires_grad for p in self.conv.parameters())))
        if grad_free and MERGE_LEVELS and len(feats) > 1 and len({x.shape[3] for x in feats}) == 1:
            # The head is shared by the levels and the predictor is pointwise: the five hidden maps are written into ONE [sum of pixels,
            # C] buffer (each 3x3 launch into its slice) and the predictor runs once over it -- one launch instead of five, four of them
            # too small to fill the chip (p3..p6: 45 + 20 + 14 + 13 us next to 132 us for p2).  Same arithmetic per pixel.
            C = feats[0].shape[3]
            ms = [x.shape[0] * x.shape[1] * x.shape[2] for x in feats]
            if MERGE_LEVELS_CONV and FUSE_PREDICTOR:
                # inference needs the hidden maps for nothing but the predictor: it runs in the 3x3 launch's epilogue
                out = K.conv3x3_levels_pred(list(feats), self.conv.packed(), pc, relu=True)
                if out is not None:
                    return out
            if sum(ms) * C * 4 < (1 << 31):
                hid = torch.empty(sum(ms), C, device=feats[0].device, dtype=torch.float32)
                views, off = [], 0
                for x, m in zip(feats, ms):
                    views.append(hid[off:off + m].view(x.shape[0], x.shape[1], x.shape[2], C))
                    off += m
                # ... and the 3x3 conv itself is one launch over the levels where its kernel allows (kernels.conv3x3_levels)
                if MERGE_LEVELS_CONV:
                    K.conv3x3_levels(list(feats), self.conv.packed(), relu=True, outs=views)
                else:
                    for x, v in zip(feats, views):
                        K.conv2d_nhwc(x, self.conv.packed(), relu=True, out=v)
                y = K.conv2d_nhwc(hid.view(1, sum(ms), 1, C), pc).view(sum(ms), -1)
                out, off = [], 0
                for x, m in zip(feats, ms):
                    out.append(y[off:off + m].view(x.shape[0], x.shape[1], x.shape[2], y.shape[1]))
                    off += m
                return out
        out = []
        for x in feats:
            h = self.conv.forward_nhwc(x)
            if train or (torch.is_grad_enabled() and h.requires_grad):
                out.append(_FusedPredictorFn.apply(h, o.weight, o.bias, d.weight, d.bias, self))
            else:
                out.append(K.conv2d_nhwc(h, pc))
        return out

    def fused_predictor_dgrad(self):
        o, d = self.objectness_logits, self.anchor_deltas
        return self._fused_dgrad.get([o.weight, d.weight],
                                     lambda: K.pack_conv_dgrad(torch.cat([o.weight, d.weight], 0), None, 0))

    def forward(self, features):
        """Reference signature: list of NCHW maps -> (list of [N,A,H,W], list of [N,A*box_dim,H,W])."""
        fused = self.forward_nhwc([to_nhwc(f) for f in features])
        A, Bd = self.num_anchors, self.box_dim
        return [to_nchw_view(t[..., :A]) for t in fused], [to_nchw_view(t[..., A:A + A * Bd]) for t in fused]


class _MatchedGT:
    """The matched gt box of every anchor, kept as (gt boxes, matched index) per image and gathered for chosen anchors only."""

    def __init__(self, per_image):
        gts = [g for g, _ in per_image]
        dev = per_image[0][1].device
        self.table = torch.cat(gts + [torch.zeros(1, 4, device=dev, dtype=gts[0].dtype)], 0)     # last row: images without gt
        n = [len(g) for g in gts]
        zero_row = sum(n)
        base, o = [], 0
        for k in n:
            base.append(o if k else zero_row)      # an image without gt has matched index 0 everywhere -> the zero row
            o += k
        self.base = torch.tensor(base, dtype=torch.int64, device=dev)
        self.midx = torch.stack([m for _, m in per_image])

    def gather(self, img, idx):
        return self.table[self.midx[img, idx] + self.base[img]]


class _FusedPredictorFn(torch.autograd.Function):
    """objectness_logits | anchor_deltas as ONE 1x1 conv over the hidden map (forward: `fused_predictor`); backward
    splits the fused weight / bias gradient back onto the two reference parameters (rpn.py:92-106)."""

    @staticmethod
    def forward(ctx, h, wo, bo, wd, bd, head):
        ctx.save_for_backward(h)
        ctx.head = head
        return K.conv2d_nhwc(h, head.fused_predictor())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        head = ctx.head
        A, Bd = head.num_anchors, head.box_dim
        g = g.contiguous()
        dw = K.conv_wgrad(h, g, None, 1, 1, 1, 0)            # [A+A*Bd (+pad), 1, 1, C]
        dw = dw.permute(0, 3, 1, 2)
        db = K.colsum_rows(g.view(-1, g.shape[-1]))
        dh = None
        if ctx.needs_input_grad[0]:
            dh = K.conv_dgrad(g[..., : A + A * Bd].contiguous(), head.fused_predictor_dgrad(), h.shape, 1)
        return dh, dw[:A].contiguous(), db[:A].contiguous(), dw[A:A + A * Bd].contiguous(), db[A:A + A * Bd].contiguous(), None


class _RpnLossFn(torch.autograd.Function):
    """RPN.losses (reference rpn.py:328-400) on the sampled anchors; the kernel emits both gradients in the forward."""

    @staticmethod
    def forward(ctx, logits, deltas, anchors, gt_boxes, labels, beta, normalizer):
        out, dl, dd = K.rpn_losses(logits, deltas, anchors, gt_boxes, labels, beta, normalizer, with_grad=True)
        ctx.save_for_backward(dl, dd)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_loc):
        dl, dd = ctx.saved_tensors
        return dl * g_cls, dd * g_loc, None, None, None, None, None


@PROPOSAL_GENERATOR_REGISTRY.register()
class RPN(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        self.in_features = cfg.MODEL.RPN.IN_FEATURES
        shapes = [input_shape[f] for f in self.in_features]
        self.rpn_head = build_rpn_head(cfg, shapes)
        self.anchor_generator = build_anchor_generator(cfg, shapes)
        self.box2box_transform = Box2BoxTransform(weights=cfg.MODEL.RPN.BBOX_REG_WEIGHTS)
        self.batch_size_per_image = cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = cfg.MODEL.RPN.POSITIVE_FRACTION
        self.pre_nms_topk = (cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN)
        self.post_nms_topk = (cfg.MODEL.RPN.POST_NMS_TOPK_TEST, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN)
        self.nms_thresh = cfg.MODEL.RPN.NMS_THRESH
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        self.anchor_boundary_thresh = cfg.MODEL.RPN.BOUNDARY_THRESH
        self.loss_weight = {"loss_rpn_cls": cfg.MODEL.RPN.LOSS_WEIGHT,
                            "loss_rpn_loc": cfg.MODEL.RPN.BBOX_REG_LOSS_WEIGHT * cfg.MODEL.RPN.LOSS_WEIGHT}
        self.box_reg_loss_type = cfg.MODEL.RPN.BBOX_REG_LOSS_TYPE
        self.smooth_l1_beta = cfg.MODEL.RPN.SMOOTH_L1_BETA
        self.anchor_matcher = Matcher(cfg.MODEL.RPN.IOU_THRESHOLDS, cfg.MODEL.RPN.IOU_LABELS, allow_low_quality_matches=True)
        assert tuple(self.box2box_transform.weights) == (1.0, 1.0, 1.0, 1.0), \
            "the fused RPN decode kernel assumes RPN.BBOX_REG_WEIGHTS == (1,1,1,1) (the default of every shipped config)"

    def predict_proposals_batched(self, feats_nhwc, image_sizes_dev, fused=None):
        """feats_nhwc: dict name -> [B,H,W,C]; image_sizes_dev: [B,2] int32 device (h,w); fused: the head's outputs
        when the caller already has them (training).  Returns (boxes [B,post,4], objectness_logits [B,post],
        count [B] int32) on device."""
        if fused is None:
            fused = self.rpn_head.forward_nhwc([feats_nhwc[f] for f in self.in_features])
        fused = [f.detach() for f in fused]
        A = self.rpn_head.num_anchors
        t = int(self.training)
        return K.rpn_proposals([f[..., :A] for f in fused], [f[..., A:5 * A] for f in fused],
                               list(self.anchor_generator.cell_anchors), self.anchor_generator.strides, image_sizes_dev,
                               self.pre_nms_topk[t], self.post_nms_topk[t], self.nms_thresh, self.min_box_size)

    # ------------------------------------------------------------------ training
    @torch.no_grad()
    def label_and_sample_anchors(self, anchors, gt_instances, lazy=False):
        """reference rpn.py:269-325.  anchors: [R,4] tensor of all anchors; returns (labels [N,R] int8 in {-1,0,1},
        matched gt boxes [N,R,4]).  lazy (the training forward): the second value is a `_MatchedGT` that gathers the gt boxes of
        the SAMPLED anchors only (`losses` needs 256 per image, not the [N,R,4] table: 34 MB per step at 268 569 anchors)."""
        gt_labels, matched_gt_boxes, matched = [], [], []
        batched = self.batched_sampling and anchors.is_cuda and len(gt_instances) > 0
        for inst in gt_instances:
            gt = inst.gt_boxes.tensor
            matched_idxs, labels = self.anchor_matcher.match(gt, anchors)
            if self.anchor_boundary_thresh >= 0:
                h, w = inst.image_size
                t = self.anchor_boundary_thresh
                inside = (anchors[:, 0] >= -t) & (anchors[:, 1] >= -t) & (anchors[:, 2] < w + t) & (anchors[:, 3] < h + t)
                labels[~inside] = -1
            if not batched:
                pos_idx, neg_idx = subsample_labels(labels, self.batch_size_per_image, self.positive_fraction, 0)
                labels.fill_(-1)
                labels.scatter_(0, pos_idx, 1)
                labels.scatter_(0, neg_idx, 0)
            gt_labels.append(labels)
            if lazy:
                matched.append((gt, matched_idxs))
            else:
                matched_gt_boxes.append(torch.zeros_like(anchors) if len(gt) == 0 else gt[matched_idxs])
        gt_labels = torch.stack(gt_labels)
        if batched:
            gt_labels = self._subsample_batched(gt_labels)
        return gt_labels, (_MatchedGT(matched) if lazy else torch.stack(matched_gt_boxes))

    batched_sampling = True     # class switch for A/B runs and tests (False: subsample_labels image by image)
    batched_targets = True      # ... True: match + sample + gather for the whole batch in five launches (csrc/train_targets.hip)

    def can_batch_targets(self, gt_instances):
        return (self.batched_targets and self.batched_sampling and self.anchor_boundary_thresh < 0 and self.box_reg_loss_type == "smooth_l1"
                and gt_instances is not None and all(len(t) <= 512 for t in gt_instances) and self.batch_size_per_image <= 1024)

    def forward_train_batched(self, images, features, gt_instances):
        """The training branch of `forward` WITHOUT its device->host read: -> (boxes [B,post,4], logits [B,post], count int32 [B], losses,
        sampled-anchor counts int32 [B,2], gt boxes [G,4], gt offsets int32 [B+1]) -- everything on the device.  The caller
        (GeneralizedRCNN._forward_train) hands the tensors to StandardROIHeads.forward_train_batched and reads all counts of the step at
        once.  Requires `can_batch_targets`."""
        feats = {f: to_nhwc(features[f]) for f in self.in_features}
        require_device(feats[self.in_features[0]], "RPN")
        sizes = self._sizes_cached(images.image_sizes, images.tensor.device)
        flist = [feats[f] for f in self.in_features]
        trains = any(p.requires_grad for p in self.parameters()) or any(f.requires_grad for f in flist)
        with torch.set_grad_enabled(trains and torch.is_grad_enabled()):
            fused = self.rpn_head.forward_nhwc(flist)
        losses, pending, (gt, gt_off) = self._losses_batched(fused, flist, gt_instances)
        with torch.no_grad():
            boxes, logits, count = self.predict_proposals_batched(feats, sizes, fused=fused)
        return boxes, logits, count, losses, pending, gt, gt_off

    def _sizes_cached(self, image_sizes, dev):
        """[B,2] int32 device tensor of the image sizes, cached by value: a fresh torch.tensor(..., device=) is a host->device copy
        from pageable memory, which waits for the work queued on the stream -- here the whole trunk."""
        cache = self.__dict__.setdefault("_sizes_cache", {})
        key = (tuple(tuple(int(v) for v in s) for s in image_sizes), str(dev))
        t = cache.get(key)
        if t is None:
            if len(cache) > 256:
                cache.clear()
            t = cache[key] = torch.tensor([list(s) for s in image_sizes], dtype=torch.int32, device=dev)
        return t

    def _anchors_cached(self, shapes, dev):
        """The grid anchors of the pyramid, concatenated, computed once per set of map shapes (they depend on nothing else)."""
        cache = self.__dict__.setdefault("_anchor_cache", {})
        key = tuple(tuple(s) for s in shapes)
        a = cache.get(key)
        if a is None or a.device != dev:
            if len(cache) > 16:
                cache.clear()
            a = cache[key] = torch.cat(self.anchor_generator._grid_anchors([tuple(s) for s in shapes]), 0).contiguous()
        return a

    def _losses_batched(self, fused, flist, gt_instances):
        """label_and_sample_anchors + losses (reference rpn.py:269-400) for the whole batch without a device->host read: one batched
        Matcher over the cached anchors, one sampling launch on the keys of ONE torch.randperm(B * R), and -- when the head does not train
        -- the sampled rows gathered straight from the per-level outputs.  -> (losses, sampled counts int32 [B,2] on the device, (gt boxes,
        gt offsets) for the ROI heads)."""
        A = self.rpn_head.num_anchors
        B = fused[0].shape[0]
        dev = fused[0].device
        anchors = self._anchors_cached([f.shape[1:3] for f in flist], dev)
        R = anchors.shape[0]
        with torch.no_grad():
            gt, gt_off, _ = K.cat_ground_truth(gt_instances)
            matches, labels = K.match_boxes_batched(gt, gt_off, B, anchors, None, self.anchor_matcher.thresholds[1:-1], self.anchor_matcher.labels, True)
            keys, seed = K.sampling_keys(B, R, dev)
            bs = self.batch_size_per_image
            sel, counts = K.subsample_batched(labels, keys, int(bs * self.positive_fraction), bs, seed=seed)
        norm = float(bs * B)
        trains = torch.is_grad_enabled() and any(f.requires_grad for f in fused)
        if not trains:
            lg, dl, an, gtb, lab = K.rpn_gather_sampled([f.detach() for f in fused], A, list(self.anchor_generator.cell_anchors),
                                                        self.anchor_generator.strides, sel, counts, matches, gt, gt_off)
            out = K.rpn_losses(lg, dl, an, gtb, lab, self.smooth_l1_beta, norm)
        else:
            # the head trains (base / ft_all yamls): rows through autograd's index path so that the gradients scatter back
            flat_logits = torch.cat([f[..., :A].reshape(B, -1) for f in fused], 1)
            flat_deltas = torch.cat([f[..., A:5 * A].reshape(B, -1, 4) for f in fused], 1)
            j = torch.arange(bs, device=dev)[None, :]
            npos, ntot = counts[:, :1].long(), counts.sum(1, keepdim=True).long()
            lab = torch.where(j < npos, 1, torch.where(j < ntot, 0, -1)).to(torch.int8).view(-1)
            idx = sel.clamp(min=0).long()
            img = torch.arange(B, device=dev)[:, None].expand(B, bs)
            midx = torch.gather(matches, 1, idx).long() + gt_off[:-1].long()[:, None]
            gtb = gt[midx.clamp(max=max(gt.shape[0] - 1, 0)).view(-1)] if gt.shape[0] else torch.zeros(B * bs, 4, device=dev)
            out = _RpnLossFn.apply(flat_logits[img, idx].view(-1), flat_deltas[img, idx].view(-1, 4), anchors[idx.view(-1)], gtb, lab,
                                   self.smooth_l1_beta, norm)
        losses = {"loss_rpn_cls": out[0] * self.loss_weight.get("loss_rpn_cls", 1.0),
                  "loss_rpn_loc": out[1] * self.loss_weight.get("loss_rpn_loc", 1.0)}
        return losses, counts, (gt, gt_off)

    def _subsample_batched(self, labels):
        """`subsample_labels` + the fill / scatter of the loop above for all images at once and without a device->host read
        (per image the loop reads the device twice and sorts ~270 000 keys twice for its two `randperm`s): ONE random permutation
        of the B*R keys, then per row the num_pos smallest keys among the positives and the num_neg smallest among the negatives
        (`topk`).  Any subset of a random permutation is in uniformly random order, so this draws the same distribution; with
        `torch.randperm` patched to arange (the parity tests) it is the reference's choice exactly: the first positives /
        negatives in anchor order."""
        B, R = labels.shape
        dev = labels.device
        bs = self.batch_size_per_image
        cap_pos = int(bs * self.positive_fraction)
        key = torch.randperm(B * R, device=dev).view(B, R)
        big = B * R
        pos, neg = labels == 1, labels == 0
        kp, kn = min(cap_pos, R), min(bs, R)
        tp = torch.topk(torch.where(pos, key, torch.full_like(key, big)), kp, dim=1, largest=False)[1]
        tn = torch.topk(torch.where(neg, key, torch.full_like(key, big)), kn, dim=1, largest=False)[1]
        npos = pos.sum(1).clamp(max=cap_pos)
        nneg = torch.minimum(neg.sum(1), bs - npos)
        out = torch.full((B, R + 1), -1, dtype=labels.dtype, device=dev)       # column R swallows the unused slots
        jp = torch.arange(kp, device=dev)[None, :]
        jn = torch.arange(kn, device=dev)[None, :]
        out.scatter_(1, torch.where(jp < npos[:, None], tp, torch.full_like(tp, R)), 1)
        out.scatter_(1, torch.where(jn < nneg[:, None], tn, torch.full_like(tn, R)), 0)
        out[:, R] = -1
        return out[:, :R].contiguous()

    def losses(self, anchors, flat_logits, gt_labels, flat_deltas, gt_boxes):
        """reference rpn.py:328-400 (smooth_l1 branch).  flat_logits [N,R], flat_deltas [N,R,4]."""
        if self.box_reg_loss_type != "smooth_l1":
            raise NotImplementedError("RPN.BBOX_REG_LOSS_TYPE '{}'".format(self.box_reg_loss_type))
        num_images = gt_labels.shape[0]
        storage = get_event_storage()
        storage.put_scalar("rpn/num_pos_anchors", float((gt_labels == 1).sum()) / num_images)
        storage.put_scalar("rpn/num_neg_anchors", float((gt_labels == 0).sum()) / num_images)
        img, idx = (gt_labels >= 0).nonzero(as_tuple=True)
        norm = float(self.batch_size_per_image * num_images)
        gtb = gt_boxes.gather(img, idx) if isinstance(gt_boxes, _MatchedGT) else gt_boxes[img, idx]
        if torch.is_grad_enabled() and flat_logits.requires_grad:
            out = _RpnLossFn.apply(flat_logits[img, idx], flat_deltas[img, idx], anchors[idx], gtb,
                                   gt_labels[img, idx], self.smooth_l1_beta, norm)
        else:
            out = K.rpn_losses(flat_logits[img, idx], flat_deltas[img, idx], anchors[idx], gtb,
                               gt_labels[img, idx], self.smooth_l1_beta, norm)
        return {"loss_rpn_cls": out[0] * self.loss_weight.get("loss_rpn_cls", 1.0),
                "loss_rpn_loc": out[1] * self.loss_weight.get("loss_rpn_loc", 1.0)}

    def forward(self, images, features, gt_instances=None):
        """Reference signature (rpn.py:402-451): -> (list[Instances], losses)."""
        feats = {f: to_nhwc(features[f]) for f in self.in_features}
        require_device(feats[self.in_features[0]], "RPN")
        sizes = self._sizes_cached(images.image_sizes, images.tensor.device)
        losses = {}
        if self.training:
            assert gt_instances is not None, "RPN requires gt_instances in training!"
            flist = [feats[f] for f in self.in_features]
            trains = any(p.requires_grad for p in self.parameters()) or any(f.requires_grad for f in flist)
            with torch.set_grad_enabled(trains and torch.is_grad_enabled()):
                fused = self.rpn_head.forward_nhwc(flist)
            A = self.rpn_head.num_anchors
            N = fused[0].shape[0]
            pending = None
            if self.can_batch_targets(gt_instances):
                losses, pending, gt_batch = self._losses_batched(fused, flist, gt_instances)
            else:
                gt_batch = None
                flat_logits = torch.cat([f[..., :A].reshape(N, -1) for f in fused], 1)
                flat_deltas = torch.cat([f[..., A:5 * A].reshape(N, -1, 4) for f in fused], 1)
                anchors = torch.cat(self.anchor_generator._grid_anchors([f.shape[1:3] for f in flist]), 0)
                gt_labels, gt_boxes = self.label_and_sample_anchors(anchors, gt_instances, lazy=True)
                losses = self.losses(anchors, flat_logits, gt_labels, flat_deltas, gt_boxes)
            with torch.no_grad():
                boxes, logits, count = self.predict_proposals_batched(feats, sizes, fused=fused)
            if pending is not None:
                # ONE device->host read for the entry point: the proposal counts and the sampled-anchor counts of the event scalars
                meta = torch.cat([count, pending.view(-1)]).tolist()
                counts = meta[:N]
                storage = get_event_storage()
                storage.put_scalar("rpn/num_pos_anchors", sum(meta[N::2]) / N)
                storage.put_scalar("rpn/num_neg_anchors", sum(meta[N + 1::2]) / N)
            else:
                counts = count.tolist()
            out = []
            for i, size in enumerate(images.image_sizes):
                inst = Instances(size)
                inst.proposal_boxes = Boxes(boxes[i, : counts[i]])
                inst.objectness_logits = logits[i, : counts[i]]
                out.append(inst)
            if gt_batch is not None and out:
                # for ROIHeads.label_and_sample_proposals: the same proposals as ONE padded table + the concatenated ground truth (a plain
                # python attribute of the first Instances, not a field: nothing else sees it)
                out[0]._lvc_batch = (boxes, logits, count, counts) + gt_batch
            return out, losses
        else:
            boxes, logits, count = self.predict_proposals_batched(feats, sizes)
        counts = count.tolist()  # the one host sync of this entry point
        out = []
        for i, size in enumerate(images.image_sizes):
            inst = Instances(size)
            inst.proposal_boxes = Boxes(boxes[i, : counts[i]])
            inst.objectness_logits = logits[i, : counts[i]]
            out.append(inst)
        return out, losses
