"""ResizeShortestEdge / ResizeTransform for uint8 images, on the device.

Mirrors the reference's test-time input transform (detectron2/data/transforms/augmentation_impl.py:184-234
`ResizeShortestEdge`, transform.py:83-134 `ResizeTransform`; built by `build_augmentation`, data/detection_utils.py:
563-595, from INPUT.MIN_SIZE_TEST / MAX_SIZE_TEST).  `apply_image` of the reference hands a uint8 HWC array to Pillow's
`Image.resize(..., BILINEAR)`; here the same fixed-point resample runs as two HIP kernels (csrc/resize.hip) on a
device tensor, bit for bit, and can write straight into the detector's normalised, padded NHWC4 batch slot.

Only the host part of Pillow's algorithm lives here: the per-output coefficient table (Resample.c precompute_coeffs
+ normalize_coeffs_8bpc for the bilinear filter over the whole image), computed in float64 with the same operation
order as the C code and cached per (input size, output size).
"""
import functools
import math

import numpy as np
import torch

from .. import kernels as K

PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=256)
def _coeffs_np(in_size, out_size):
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size     # Pillow keeps the box in float
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                                         # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), np.float64)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):                                              # ww accumulates tap by tap, as the C loop does
        v = np.abs(((x + xmin).astype(np.float64) - center + 0.5) * ss)
        w = np.where(v < 1.0, 1.0 - v, 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    p = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, np.trunc(-0.5 + p), np.trunc(0.5 + p)).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk, ksize


_DEV_COEFFS = {}


def resample_coeffs(in_size, out_size, device=None):
    """(bounds [out,2] int32, coefficients [out,ksize] int32, ksize); on `device` when given (cached)."""
    b, k, ks = _coeffs_np(int(in_size), int(out_size))
    if device is None:
        return b, k, ks
    key = (int(in_size), int(out_size), str(device))
    hit = _DEV_COEFFS.get(key)
    if hit is None:
        hit = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
        _DEV_COEFFS[key] = hit
    return hit


class ResizeTransform:
    """reference transform.py:83-134 for uint8 images and coordinates."""

    def __init__(self, h, w, new_h, new_w, interp=None):
        if interp not in (None, 2, "bilinear"):     # PIL.Image.BILINEAR == 2
            raise NotImplementedError("only the bilinear resize of the shipped configs is implemented")
        self.h, self.w, self.new_h, self.new_w = int(h), int(w), int(new_h), int(new_w)

    def apply_image(self, img, out_slot=None, mean=None, std=None):
        """img: uint8 [H,W,3] tensor (moved to the device if it is not there).  Returns the resized uint8 [new_h,new_w,3]
        device tensor; with `out_slot` ([Hp,Wp,4] fp32 view of the batch) also writes (resized - mean) / std zero-padded
        into it (GeneralizedRCNN.preprocess_image fused in)."""
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
        assert tuple(img.shape[:2]) == (self.h, self.w)
        return K.resize_bilinear_u8(img, self.new_h, self.new_w, resample_coeffs, out_slot=out_slot, mean=mean, std=std)

    def apply_coords(self, coords):
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords

    def apply_box(self, box):
        box = box.clone().reshape(-1, 4)
        box[:, 0::2] = box[:, 0::2] * (self.new_w * 1.0 / self.w)
        box[:, 1::2] = box[:, 1::2] * (self.new_h * 1.0 / self.h)
        return box

    def inverse(self):
        return ResizeTransform(self.new_h, self.new_w, self.h, self.w)


class ResizeShortestEdge:
    """reference augmentation_impl.py:184-234.  Test time uses sample_style "choice" with one length
    (detection_utils.py:577-581), so no random draw is involved on the path; "range" draws with numpy as the
    reference does."""

    def __init__(self, short_edge_length, max_size=2 ** 63 - 1, sample_style="range", interp=None):
        assert sample_style in ["range", "choice"], sample_style
        self.is_range = sample_style == "range"
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        if self.is_range:
            assert len(short_edge_length) == 2, "short_edge_length must be two values using 'range' sample style."
        self.short_edge_length, self.max_size, self.interp = tuple(short_edge_length), max_size, interp

    @classmethod
    def from_config(cls, cfg, is_train=False):
        """build_augmentation (detection_utils.py:563-595)."""
        I = cfg.INPUT
        if is_train:
            return cls(I.MIN_SIZE_TRAIN, I.MAX_SIZE_TRAIN, I.MIN_SIZE_TRAIN_SAMPLING)
        return cls(I.MIN_SIZE_TEST, I.MAX_SIZE_TEST, "choice")

    def output_size(self, h, w, size):
        scale = size * 1.0 / min(h, w)
        if h < w:
            newh, neww = size, scale * w
        else:
            newh, neww = scale * h, size
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh = newh * scale
            neww = neww * scale
        return int(newh + 0.5), int(neww + 0.5)

    def get_transform(self, img):
        h, w = int(img.shape[0]), int(img.shape[1])
        if self.is_range:
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return None     # NoOpTransform
        newh, neww = self.output_size(h, w, int(size))
        return ResizeTransform(h, w, newh, neww, self.interp)
