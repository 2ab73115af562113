"""PyTorch custom ops `torch.ops.lvc_amd.*`: the reference's only native seam, re-exported with the same schemas.

  reference                                                             here
  _C.roi_align_forward  (detectron2/layers/csrc/vision.cpp:96,          torch.ops.lvc_amd.roi_align_forward
                         ROIAlign/ROIAlign.h:54-86)
  _C.roi_align_backward (vision.cpp:97, ROIAlign.h:88-128)              torch.ops.lvc_amd.roi_align_backward
  torchvision.ops.nms          (detectron2/layers/nms.py:7)             torch.ops.lvc_amd.nms
  torchvision.ops.boxes.batched_nms (detectron2/layers/nms.py:6, 20)    torch.ops.lvc_amd.batched_nms

Positional signatures, output shapes / dtypes and ownership are the reference's: outputs are fresh tensors, inputs are
made contiguous internally and never mutated, launches go to the current stream and nothing synchronises (the
reference's `cudaDeviceSynchronize` in ROIAlign_cuda.cu:364 is a defect there).  `roi_align_forward` carries the
autograd formula of the reference's `_ROIAlign` (detectron2/layers/roi_align.py:22-57: gradient for `input` only,
through `roi_align_backward`).  The implementations are the C-ABI kernels of liblvc_amd.so (lvc_amd/kernels.py); CPU
tensors raise RuntimeError -- there is no CPU path.  Fake (meta) kernels give shapes to torch.export / torch.compile.
"""
from typing import Optional

import torch
from torch import Tensor

from . import kernels as K

_lib_def = torch.library.Library("lvc_amd", "DEF")
_lib_def.define("roi_align_forward(Tensor input, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, "
                "int sampling_ratio, bool aligned) -> Tensor")
_lib_def.define("roi_align_backward(Tensor grad, Tensor rois, float spatial_scale, int pooled_height, int pooled_width, "
                "int batch_size, int channels, int height, int width, int sampling_ratio, bool aligned) -> Tensor")
_lib_def.define("nms(Tensor boxes, Tensor scores, float iou_threshold) -> Tensor")
_lib_def.define("batched_nms(Tensor boxes, Tensor scores, Tensor idxs, float iou_threshold) -> Tensor")


def _fwd(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned):
    return K.roi_align_forward(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned)


def _bwd(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio, aligned):
    return K.roi_align_backward(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height,
                                width, sampling_ratio, aligned)


def _nms(boxes, scores, iou_threshold):
    return K.nms(boxes, scores, iou_threshold)


def _batched_nms(boxes, scores, idxs, iou_threshold):
    return K.batched_nms(boxes, scores, idxs, iou_threshold)


# CompositeExplicitAutograd: one implementation for every backend key; the kernels raise RuntimeError for CPU tensors
# themselves (the reference's AT_ERROR("Not compiled with GPU support") analogue)
_lib_def.impl("roi_align_forward", _fwd, "CompositeExplicitAutograd")
_lib_def.impl("roi_align_backward", _bwd, "CompositeExplicitAutograd")
_lib_def.impl("nms", _nms, "CompositeExplicitAutograd")
_lib_def.impl("batched_nms", _batched_nms, "CompositeExplicitAutograd")


@torch.library.register_fake("lvc_amd::roi_align_forward")
def _(input, rois, spatial_scale, pooled_height, pooled_width, sampling_ratio, aligned):
    return input.new_empty((rois.shape[0], input.shape[1], pooled_height, pooled_width), dtype=torch.float32)


@torch.library.register_fake("lvc_amd::roi_align_backward")
def _(grad, rois, spatial_scale, pooled_height, pooled_width, batch_size, channels, height, width, sampling_ratio, aligned):
    return grad.new_empty((batch_size, channels, height, width), dtype=torch.float32)


@torch.library.register_fake("lvc_amd::nms")
def _(boxes, scores, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


@torch.library.register_fake("lvc_amd::batched_nms")
def _(boxes, scores, idxs, iou_threshold):
    n = torch.library.get_ctx().new_dynamic_size()
    return boxes.new_empty((n,), dtype=torch.int64)


def _roi_align_setup(ctx, inputs, output):
    input, rois, spatial_scale, ph, pw, sampling_ratio, aligned = inputs
    ctx.save_for_backward(rois)
    ctx.args = (spatial_scale, ph, pw, tuple(input.shape), sampling_ratio, aligned)


def _roi_align_backward(ctx, grad_output):
    (rois,) = ctx.saved_tensors
    spatial_scale, ph, pw, (bs, ch, h, w), sampling_ratio, aligned = ctx.args
    gin = torch.ops.lvc_amd.roi_align_backward(grad_output, rois, spatial_scale, ph, pw, bs, ch, h, w, sampling_ratio, aligned)
    return gin, None, None, None, None, None, None


torch.library.register_autograd("lvc_amd::roi_align_forward", _roi_align_backward, setup_context=_roi_align_setup)

roi_align_forward = torch.ops.lvc_amd.roi_align_forward
roi_align_backward = torch.ops.lvc_amd.roi_align_backward
nms = torch.ops.lvc_amd.nms
batched_nms = torch.ops.lvc_amd.batched_nms
