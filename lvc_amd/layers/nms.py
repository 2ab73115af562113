"""`nms` / `batched_nms` with the reference's signatures (detectron2/layers/nms.py:6-29): the custom ops
`torch.ops.lvc_amd.nms` / `torch.ops.lvc_amd.batched_nms` on the gfx950 ballot kernels (bit-exact keep indices w.r.t.
the CPU algorithm; see csrc/nms.hip)."""
import torch

from .. import ops


def nms(boxes, scores, iou_threshold):
    return ops.nms(boxes, scores, float(iou_threshold))


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Same as torchvision.ops.boxes.batched_nms, but safer (reference nms.py:10-29): fp16 boxes are upcast."""
    assert boxes.shape[-1] == 4
    return ops.batched_nms(boxes.float(), scores, idxs, float(iou_threshold))
