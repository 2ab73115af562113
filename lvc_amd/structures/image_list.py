"""`ImageList`: a batch of differently sized images padded into one tensor.

Contract restated from reference detectron2/structures/image_list.py:8-119: `.tensor` is logically
[N,C,Hmax,Wmax] zero padded bottom/right to a multiple of `size_divisibility`, `.image_sizes` the
un-padded (h, w) list, `len`, `[i]` returns the un-padded view, `.to`, `.device`.

MI355X-first difference: tensors produced by `GeneralizedRCNN.preprocess_image` are physically NHWC
with 4 channel slots (channels-last, 16-byte pixels) so the stem kernel can load one pixel per lane
with a single dwordx4; `.tensor` is an NCHW-shaped *view* of that storage, so shape semantics at the
boundary are unchanged.
"""
import torch
import torch.nn.functional as F


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    def to(self, *args, **kwargs):
        return ImageList(self.tensor.to(*args, **kwargs), self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def padded_size(sizes, size_divisibility=0):
        mh = max(s[0] for s in sizes)
        mw = max(s[1] for s in sizes)
        if size_divisibility > 1:
            d = size_divisibility
            mh = (mh + d - 1) // d * d
            mw = (mw + d - 1) // d * d
        return mh, mw

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        """Generic (torch) path, for callers that build an ImageList themselves."""
        assert len(tensors) > 0
        assert isinstance(tensors, (tuple, list))
        for t in tensors:
            assert isinstance(t, torch.Tensor), type(t)
            assert t.shape[1:-2] == tensors[0].shape[1:-2], t.shape
        image_sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in tensors]
        mh, mw = ImageList.padded_size(image_sizes, size_divisibility)
        if len(tensors) == 1:
            h, w = image_sizes[0]
            batched = F.pad(tensors[0], [0, mw - w, 0, mh - h], value=pad_value).unsqueeze_(0)
        else:
            shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [mh, mw]
            batched = tensors[0].new_full(shape, pad_value)
            for img, pad_img in zip(tensors, batched):
                pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)
