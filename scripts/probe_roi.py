"""ROIAlign of a real step (8 x 1000 proposals of the bench batch on p2..p5), event-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
    out = pooler.pool_nhwc(flist, pboxes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = pooler.pool_nhwc(flist, pboxes)
    e1.record(); torch.cuda.synchronize()
    wh = (pboxes[..., 2:] - pboxes[..., :2]).reshape(-1, 2)
    print("%.4f ms  checksum %.6e   median proposal %.0f x %.0f px" % (e0.elapsed_time(e1) / 20, float(out.double().sum()), float(wh[:, 0].median()), float(wh[:, 1].median())))
