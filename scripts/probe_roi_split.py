"""ROIAlign of the bench batch's proposals by size class: the RoIs sorted by window area (feature pixels at their level), timed in
deciles -- is the launch bound by its many small items or by its few large ones?"""
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
    pooler = model.roi_heads.box_pooler
    levels, rois = K.assign_levels_rois(pboxes, pooler.min_level, pooler.max_level, pooler.canonical_box_size, pooler.canonical_level)
    sc = torch.tensor(pooler.scales, device=dev)[levels.long()]
    w = (rois[:, 3] - rois[:, 1]) * sc
    h = (rois[:, 4] - rois[:, 2]) * sc
    area = (w + 2) * (h + 2 * 7)          # window pixels summed over the seven output rows, roughly
    order = torch.argsort(area)

    def timed(r, l, n=20):
        for _ in range(3):
            K.roi_align_fpn_nhwc(flist, pooler.scales, r, l, 7, 7, pooler.sampling_ratio, pooler.aligned)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            K.roi_align_fpn_nhwc(flist, pooler.scales, r, l, 7, 7, pooler.sampling_ratio, pooler.aligned)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    print("all %d RoIs: %.4f ms; same RoIs sorted by size: %.4f ms" % (len(rois), timed(rois, levels), timed(rois[order].contiguous(), levels[order].contiguous())))
    desc = torch.flip(order, [0])
    print("sorted descending: %.4f ms" % timed(rois[desc].contiguous(), levels[desc].contiguous()))
    bucket = torch.log2(area.clamp(min=1)).floor()
    for name, key in (("bucket log2(area) ascending (stable)", bucket), ("bucket log2(area) descending (stable)", -bucket),
                      ("half-octave buckets descending", -(2 * torch.log2(area.clamp(min=1))).floor())):
        o = torch.sort(key, stable=True)[1]
        print("%s: %.4f ms" % (name, timed(rois[o].contiguous(), levels[o].contiguous())))
    print("all again: %.4f ms" % timed(rois, levels))
    K_ = len(rois)
    for d in range(0):
        sel = order[d * K_ // 10:(d + 1) * K_ // 10]
        r, l = rois[sel].contiguous(), levels[sel].contiguous()
        print("decile %d: window area %6.0f..%6.0f px  %.4f ms" % (d, float(area[sel].min()), float(area[sel].max()), timed(r, l)))
    sel = order[: 9 * K_ // 10]
    print("smallest 90 %%: %.4f ms" % timed(rois[sel].contiguous(), levels[sel].contiguous()))
