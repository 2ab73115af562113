"""ROIAlign of the bench batch's proposals under different work orders (which workgroup -- and with block i on XCD i % 8, which
XCD -- pools which RoI): largest first (the built-in order), image b on XCD b, spatial order inside an image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lvc_amd import kernels as K
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn
model = build_model(base_rcnn_fpn()).eval(); syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]
with torch.no_grad():
    images = model.preprocess_image(batch)
    sizes_dev = model._dev_const(images.image_sizes, torch.int32)
    N, _, Hp, Wp = images.tensor.shape
    x4 = images.tensor.as_strided((N, Hp, Wp, 4), (Hp * Wp * 4, Wp * 4, 4, 1), images.tensor.storage_offset())
    feats = model.backbone.forward_nhwc(x4)
    pboxes, _, _ = model.proposal_generator.predict_proposals_batched(feats, sizes_dev)
    flist = [feats[f] for f in model.roi_heads.in_features]
    P = model.roi_heads.box_pooler
    levels, rois = K.assign_levels_rois(pboxes, P.min_level, P.max_level, P.canonical_box_size, P.canonical_level)
    sc = torch.tensor(P.scales, device=dev)[levels.long()]
    area = ((rois[:, 3] - rois[:, 1]) * sc + 2) * ((rois[:, 4] - rois[:, 2]) * sc + 14)
    img = rois[:, 0].long()
    yc = (rois[:, 2] + rois[:, 4]) * 0.5
    Kn = len(rois)

    def timed(order, n=20):
        o = None if order is None else order.to(torch.int32).contiguous()
        for _ in range(3): K.roi_align_fpn_nhwc(flist, P.scales, rois, levels, 7, 7, P.sampling_ratio, P.aligned, order=o)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): K.roi_align_fpn_nhwc(flist, P.scales, rois, levels, 7, 7, P.sampling_ratio, P.aligned, order=o)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def interleave(keys):
        """position i holds a RoI of image i % 8 (block i runs on XCD i % 8); inside an image ascending `keys`"""
        per = []
        for b in range(8):
            idx = (img == b).nonzero().view(-1)
            per.append(idx[torch.argsort(keys[idx])])
        m = min(len(p) for p in per)
        head = torch.stack([p[:m] for p in per], 1).reshape(-1)
        tail = torch.cat([p[m:] for p in per])
        return torch.cat([head, tail])

    bucket = (2 * torch.log2(area.clamp(min=1))).floor()
    print("built-in order (largest window first): %.4f ms" % timed(None))
    print("proposal order: %.4f ms" % timed(torch.arange(Kn, device=dev)))
    print("image b on XCD b, largest first inside: %.4f ms" % timed(interleave(-area)))
    print("image b on XCD b, by centre row inside: %.4f ms" % timed(interleave(yc)))
    print("image b on XCD b, area buckets (largest first), centre row inside a bucket: %.4f ms" % timed(interleave(-bucket * 4096 + yc)))
    lv = levels.float()
    print("image b on XCD b, level, then centre row: %.4f ms" % timed(interleave(lv * 4096 + yc)))
    print("built-in again: %.4f ms" % timed(None))
