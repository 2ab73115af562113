"""`Boxes`: the [N,4] fp32 XYXY box container that crosses the drop-in boundary.

Contract restated from reference detectron2/structures/boxes.py:132-312 (the callers of the hot
path read `.tensor`, `len()`, indexing, `area`, in-place `clip`/`scale`, `nonempty` (strict `>`),
`Boxes.cat`, `.to`, `.device`) and :315-347 (`pairwise_iou`).  These are small host-side helpers
for callers; the hot path itself computes boxes inside the HIP kernels (csrc/boxes.hip).
"""
from enum import IntEnum, unique

import torch


@unique
class BoxMode(IntEnum):
    """reference boxes.py:13-130 (only the axis-aligned conversions the detector's callers use)."""

    XYXY_ABS = 0
    XYWH_ABS = 1
    XYXY_REL = 2
    XYWH_REL = 3
    XYWHA_ABS = 4

    @staticmethod
    def convert(box, from_mode, to_mode):
        if from_mode == to_mode:
            return box
        original_type = type(box)
        is_numpy = False
        single = isinstance(box, (list, tuple))
        if single:
            arr = torch.tensor(box, dtype=torch.float64)[None, :]
        elif isinstance(box, torch.Tensor):
            arr = box.clone()
        else:
            import numpy as np

            is_numpy = isinstance(box, np.ndarray)
            arr = torch.from_numpy(np.asarray(box)).clone()
        pair = (from_mode, to_mode)
        if pair == (BoxMode.XYWH_ABS, BoxMode.XYXY_ABS):
            arr[:, 2] += arr[:, 0]
            arr[:, 3] += arr[:, 1]
        elif pair == (BoxMode.XYXY_ABS, BoxMode.XYWH_ABS):
            arr[:, 2] -= arr[:, 0]
            arr[:, 3] -= arr[:, 1]
        else:
            raise NotImplementedError("Conversion from BoxMode {} to {} is not supported".format(from_mode, to_mode))
        if single:
            return original_type(arr.flatten().tolist())
        if is_numpy:
            return arr.numpy()
        return arr


class Boxes:
    def __init__(self, tensor):
        device = tensor.device if isinstance(tensor, torch.Tensor) else torch.device("cpu")
        tensor = torch.as_tensor(tensor, dtype=torch.float32, device=device)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, 4)).to(dtype=torch.float32, device=device)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, *args, **kwargs):
        return Boxes(self.tensor.to(*args, **kwargs))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        assert torch.isfinite(self.tensor).all(), "Box tensor contains infinite or NaN!"
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def inside_box(self, box_size, boundary_threshold=0):
        h, w = box_size
        t = self.tensor
        return ((t[..., 0] >= -boundary_threshold) & (t[..., 1] >= -boundary_threshold)
                & (t[..., 2] < w + boundary_threshold) & (t[..., 3] < h + boundary_threshold))

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    def scale(self, scale_x, scale_y):
        self.tensor[:, 0::2] *= scale_x
        self.tensor[:, 1::2] *= scale_y

    @classmethod
    def cat(cls, boxes_list):
        assert isinstance(boxes_list, (list, tuple))
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        assert all(isinstance(b, Boxes) for b in boxes_list)
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1, boxes2):
    """IoU matrix [N,M] (reference boxes.py:315-347: intersection 0 where the boxes do not overlap,
    IoU 0 where the intersection is 0)."""
    a1, a2 = boxes1.area(), boxes2.area()
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype, device=inter.device))
