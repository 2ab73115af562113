"""RBG: random-box generator around the ground truth for the box corrector (reference
lvc/modeling/proposal_generator/rbg.py:10-160).  Evaluation passes the loaded proposals through unchanged (:49-50);
training keeps loaded proposals whose best IoU with a GT exceeds `t` and adds `2*positive_num/len(gt)` jittered
copies of every GT box (centre shift ~U(-alpha,alpha)*size, log-size ~U(log(1-beta), log(1+beta))), clipped, kept if
IoU > t.  Pure elementwise + RNG (torch's generator, as in the reference); the IoU filter runs on the match kernel.
"""
import math

import torch
from torch import nn

from ... import kernels as K
from ...structures import Boxes, Instances
from .build import PROPOSAL_GENERATOR_REGISTRY


@PROPOSAL_GENERATOR_REGISTRY.register()
class RBG(nn.Module):
    def __init__(self, cfg, input_shape=None):
        super().__init__()
        self.alpha = cfg.MODEL.RBG.ALPHA
        self.beta = cfg.MODEL.RBG.BETA
        self.t = cfg.MODEL.RBG.T
        self.positive_num_per_image = int(cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE * cfg.MODEL.ROI_HEADS.POSITIVE_FRACTION)

    def _max_iou(self, gt, boxes):
        if boxes.is_cuda:
            return K.match_boxes(gt, boxes, [0.5], [0, 1], False)[2]
        from ...structures import pairwise_iou

        return pairwise_iou(Boxes(gt), Boxes(boxes)).max(dim=0)[0]

    def forward(self, proposals, targets):
        if not self.training:
            return proposals, {}
        filtered = self.filter_proposals(proposals, targets)
        new = self.generate_proposals(targets)
        return [Instances.cat([f, n]) for f, n in zip(filtered, new)], {}

    batched = True      # class switch for A/B runs and tests (False: the per-image lists of `forward`)

    def can_batch(self, proposals, targets):
        return (self.batched and self.training and targets is not None and len(targets) > 0 and len(targets) == len(proposals)
                and all(0 < len(t) <= 512 and t.gt_boxes.tensor.is_cuda for t in targets))

    @torch.no_grad()
    def forward_batched(self, proposals, targets, gt, gt_off):
        """`forward` for the whole batch without a device->host read: the loaded proposals followed by the jittered ground-truth
        boxes of every image as ONE padded table [B,P,4], and instead of dropping the rows whose best IoU is <= T (a boolean index =
        a read per image) a mask of the rows that stay.  Row order inside an image = the reference's list with gaps (reference
        lvc/modeling/proposal_generator/rbg.py: filter_proposals, generate_proposals).  -> (boxes [B,P,4], keep bool [B,P])."""
        dev = gt.device
        per = []
        for tar, prop in zip(targets, proposals):
            g = tar.gt_boxes.tensor
            G = len(tar)
            N = (2 * self.positive_num_per_image) // G
            lo, hi = math.log(1 - self.beta), math.log(1 + self.beta)
            tx = torch.ones((N, G), device=dev).uniform_(-self.alpha, self.alpha)
            ty = torch.ones((N, G), device=dev).uniform_(-self.alpha, self.alpha)
            tw = torch.ones((N, G), device=dev).uniform_(lo, hi)
            th = torch.ones((N, G), device=dev).uniform_(lo, hi)
            x, y = g[:, 0:1], g[:, 1:2]
            w, h = g[:, 2:3] - g[:, 0:1], g[:, 3:4] - g[:, 1:2]
            nx, ny = x + w * tx.t(), y + h * ty.t()
            nw, nh = w * torch.exp(tw.t()), h * torch.exp(th.t())
            xyxy = torch.stack([nx, ny, nx + nw, ny + nh]).permute(1, 2, 0).reshape(-1, 4)
            boxes = Boxes(xyxy)
            boxes.clip(tar.image_size)
            per.append(torch.cat([prop.proposal_boxes.tensor.to(dev), boxes.tensor], 0))
        B = len(per)
        P = max(max(len(p) for p in per), 1)
        table = torch.zeros(B, P, 4, device=dev)
        rows = torch.zeros(B, dtype=torch.int64)
        for i, p in enumerate(per):
            table[i, : len(p)] = p
            rows[i] = len(p)
        _m, _l, vals = K.match_boxes_batched(gt, gt_off, B, table, None, [0.5], [0, 1], False, return_vals=True)
        keep = (vals > self.t) & (torch.arange(P, device=dev)[None, :] < rows.pin_memory().to(dev, non_blocking=True)[:, None])
        return table, keep

    def filter_proposals(self, proposals, targets):
        out = []
        for tar, prop in zip(targets, proposals):
            gt, pb = tar.gt_boxes.tensor, prop.proposal_boxes.tensor
            if len(pb) and len(gt):
                out.append(prop[self._max_iou(gt, pb) > self.t])
            else:
                out.append(Instances(tar.image_size, proposal_boxes=Boxes(torch.zeros((0, 4)).to(gt)),
                                     objectness_logits=torch.zeros((0), device=gt.device)))
        return out

    def generate_proposals(self, targets):
        out = []
        for tar in targets:
            gt = tar.gt_boxes.tensor
            dev = gt.device
            G = len(tar)
            if G == 0:
                out.append(Instances(tar.image_size, proposal_boxes=Boxes(torch.zeros(0, 4).to(dev)),
                                     objectness_logits=torch.zeros(0).to(dev)))
                continue
            N = (2 * self.positive_num_per_image) // G
            lo, hi = math.log(1 - self.beta), math.log(1 + self.beta)
            tx = torch.ones((N, G), device=dev).uniform_(-self.alpha, self.alpha)
            ty = torch.ones((N, G), device=dev).uniform_(-self.alpha, self.alpha)
            tw = torch.ones((N, G), device=dev).uniform_(lo, hi)
            th = torch.ones((N, G), device=dev).uniform_(lo, hi)
            x, y = gt[:, 0:1], gt[:, 1:2]
            w, h = gt[:, 2:3] - gt[:, 0:1], gt[:, 3:4] - gt[:, 1:2]
            nx, ny = x + w * tx.t(), y + h * ty.t()
            nw, nh = w * torch.exp(tw.t()), h * torch.exp(th.t())
            xyxy = torch.stack([nx, ny, nx + nw, ny + nh]).permute(1, 2, 0).reshape(-1, 4)
            boxes = Boxes(xyxy)
            boxes.clip(tar.image_size)
            boxes = boxes[self._max_iou(gt, boxes.tensor) > self.t]
            out.append(Instances(tar.image_size, proposal_boxes=boxes, objectness_logits=torch.ones(len(boxes), device=dev)))
        return out
