"""Matcher (reference detectron2/modeling/matcher.py:8-126): the same constructor contract; `match(gt, boxes)`
runs pairwise_iou + thresholding + low-quality matches as one two-pass kernel (csrc/train.hip) without
materialising the G x N IoU matrix (268 569 anchors x G for the RPN)."""
import torch

from .. import kernels as K


class Matcher:
    def __init__(self, thresholds, labels, allow_low_quality_matches=False):
        thresholds = list(thresholds[:])
        assert thresholds[0] > 0
        assert all(lo <= hi for lo, hi in zip(thresholds[:-1], thresholds[1:]))
        assert all(l in [-1, 0, 1] for l in labels)
        assert len(labels) == len(thresholds) + 1
        if len(thresholds) not in (1, 2):
            raise NotImplementedError("Matcher supports 1 or 2 IoU thresholds (all shipped configs)")
        self.user_thresholds = thresholds
        self.thresholds = [-float("inf")] + thresholds + [float("inf")]
        self.labels = list(labels)
        self.allow_low_quality_matches = allow_low_quality_matches

    def match(self, gt_boxes, boxes):
        """gt_boxes [G,4], boxes [N,4] device tensors -> (matches int64 [N], labels int8 [N])."""
        N = boxes.shape[0]
        if gt_boxes.shape[0] == 0:  # reference matcher.py:76-84
            return (torch.zeros(N, dtype=torch.int64, device=boxes.device),
                    torch.full((N,), self.labels[0], dtype=torch.int8, device=boxes.device))
        m, l, _ = K.match_boxes(gt_boxes, boxes, self.user_thresholds, self.labels, self.allow_low_quality_matches)
        return m, l
