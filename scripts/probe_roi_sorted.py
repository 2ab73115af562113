"""ROIAlign of a real step with the RoIs handed over in spatial order (level, image, 64-px cell of the centre) against the
proposal (score) order: kernel time only; the XCD-local row mapping deals RoIs to XCDs round robin, so sorting cannot pile a
level onto one XCD."""
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

    def run(r, l):
        return K.roi_align_fpn_nhwc(flist, P.scales, r, l, 7, 7, P.sampling_ratio, P.aligned)

    def timeit(r, l):
        for _ in range(3): run(r, l)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(r, l)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20

    cx = (rois[:, 1] + rois[:, 3]) * 0.5
    cy = (rois[:, 2] + rois[:, 4]) * 0.5
    for cell in (32, 64, 128):
        key = ((levels.long() * 16 + rois[:, 0].long()) * 64 + (cy / cell).long().clamp(0, 63)) * 64 + (cx / cell).long().clamp(0, 63)
        perm = torch.argsort(key)
        ts = timeit(rois[perm].contiguous(), levels[perm].contiguous())
        print("sorted by %3d-px cell: %.4f ms" % (cell, ts))
    print("proposal order:        %.4f ms" % timeit(rois, levels))
    a = run(rois, levels); b = run(rois[perm].contiguous(), levels[perm].contiguous())
    print("same values:", torch.equal(a[perm], b))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        key = ((levels.long() * 16 + rois[:, 0].long()) * 64 + (cy / 64).long().clamp(0, 63)) * 64 + (cx / 64).long().clamp(0, 63)
        perm = torch.argsort(key)
    e1.record(); torch.cuda.synchronize()
    print("key + argsort (torch): %.4f ms" % (e0.elapsed_time(e1) / 20))
