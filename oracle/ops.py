"""ctypes bindings of oracle/liboracle.so + torch restatements of the small third-party box ops.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "oracle.c")
        inc = os.path.join(_HERE, "roi_align_fwd.inc")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(inc)):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
        L = ctypes.CDLL(so)
        L.orc_roi_align_forward.restype = ctypes.c_int
        L.orc_roi_align_forward_f64.restype = ctypes.c_int
        L.orc_roi_align_backward.restype = ctypes.c_int
        L.orc_nms.restype = ctypes.c_int64
        _LIB = L
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned):
    """Same signature as the reference `_C.roi_align_forward` (csrc/vision.cpp:96)."""
    if input.dtype == torch.float64:   # ROIAlignForward<double>: the fp64 evaluation of the path
        input = input.contiguous()
        rois = rois.contiguous().double()
        K = rois.shape[0]
        B, C, H, W = input.shape
        out = torch.zeros(K, C, pooled_h, pooled_w, dtype=torch.float64)
        rc = lib().orc_roi_align_forward_f64(_p(input), _p(rois), _p(out), K, C, H, W, pooled_h, pooled_w,
                                             ctypes.c_double(spatial_scale), sampling_ratio, int(bool(aligned)))
        if rc != 0:
            raise RuntimeError("ROIs in ROIAlign cannot have non-negative size!")
        return out
    input = input.contiguous().float()
    rois = rois.contiguous().float()
    K = rois.shape[0]
    B, C, H, W = input.shape
    out = torch.zeros(K, C, pooled_h, pooled_w, dtype=torch.float32)
    rc = lib().orc_roi_align_forward(_p(input), _p(rois), _p(out), K, C, H, W, pooled_h, pooled_w,
                                     ctypes.c_float(spatial_scale), sampling_ratio, int(bool(aligned)))
    if rc != 0:
        raise RuntimeError("ROIs in ROIAlign cannot have non-negative size!")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_h, pooled_w, batch_size, channels, height, width,
                       sampling_ratio, aligned):
    """Same signature as the reference `_C.roi_align_backward` (csrc/vision.cpp:97, ROIAlign.h:88-128)."""
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    K = rois.shape[0]
    gin = torch.zeros(batch_size, channels, height, width, dtype=torch.float32)
    rc = lib().orc_roi_align_backward(_p(grad), _p(rois), _p(gin), K, batch_size, channels, height, width,
                                      pooled_h, pooled_w, ctypes.c_float(spatial_scale), sampling_ratio,
                                      int(bool(aligned)))
    if rc != 0:
        raise RuntimeError("ROIs in ROIAlign do not have non-negative size!")
    return gin


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms (0.8.2 CPU algorithm): keep indices, score-descending."""
    boxes = boxes.contiguous().float()
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64)
    # The reference sorts with an UNSTABLE sort (torchvision nms_cpu.cpp: scores.sort(0, true)), so the
    # order of exactly tied scores is unspecified there; the oracle (and the HIP kernels) fix it to
    # "lower index first" = torch's stable sort.
    order = scores.sort(dim=0, descending=True, stable=True)[1].contiguous()
    keep = torch.empty(n, dtype=torch.int64)
    nk = lib().orc_nms(_p(boxes), _p(order), ctypes.c_int64(n), ctypes.c_double(iou_threshold), _p(keep))
    return keep[:nk].clone()


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.boxes.batched_nms (0.8.2): offset boxes by idx*(max_coordinate+1)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    return nms(boxes_for_nms, scores, iou_threshold)


def box_iou(boxes1, boxes2):
    """torchvision.ops.box_iou (0.8.2)."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area1[:, None] + area2 - inter)
