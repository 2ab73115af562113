"""Kernel-level timing of the non-conv stages for one bench batch (GPU only)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvc_amd import kernels as K
from lvc_amd.config.presets import base_rcnn_fpn
from lvc_amd.modeling import build_model
from lvc_amd.utils import synthetic as syn

model = build_model(base_rcnn_fpn()).eval()
syn.conditioned_r50_fpn_(model)
dev = torch.device("cuda:0")
batch = [{"image": syn.synthetic_image(1 + i).to(dev), "height": 800, "width": 1333} for i in range(8)]

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

with torch.no_grad():
    images = model.preprocess_image(batch)
    sizes_dev = torch.tensor([list(s) for s in images.image_sizes], dtype=torch.int32, device=dev)
    N, _, Hp, Wp = images.tensor.shape
    x4 = images.tensor.as_strided((N, Hp, Wp, 4), (Hp * Wp * 4, Wp * 4, 4, 1), images.tensor.storage_offset())
    feats = model.backbone.forward_nhwc(x4)
    pg, rh = model.proposal_generator, model.roi_heads
    pboxes, plog, pcount = pg.predict_proposals_batched(feats, sizes_dev)
    fl = [feats[f] for f in rh.in_features]
    print("preprocess        %.3f ms" % timeit(lambda: model.preprocess_image(batch)))
    print("backbone          %.3f ms" % timeit(lambda: model.backbone.forward_nhwc(x4)))
    print("rpn total         %.3f ms" % timeit(lambda: pg.predict_proposals_batched(feats, sizes_dev)))
    fused = pg.rpn_head.forward_nhwc([feats[f] for f in pg.in_features])
    print("  rpn head convs  %.3f ms" % timeit(lambda: pg.rpn_head.forward_nhwc([feats[f] for f in pg.in_features])))
    print("roi pool          %.3f ms" % timeit(lambda: rh.box_pooler.pool_nhwc(fl, pboxes)))
    pooled = rh.box_pooler.pool_nhwc(fl, pboxes)
    print("box head          %.3f ms" % timeit(lambda: rh.box_head.forward_nhwc(pooled)))
    h = rh.box_head.forward_nhwc(pooled)
    print("predictor         %.3f ms" % timeit(lambda: rh.box_predictor(h)))
    sc, dl = rh.box_predictor(h)
    st = K.new_status(dev)
    print("det inference     %.3f ms" % timeit(lambda: K.fast_rcnn_inference(sc, dl, pboxes, pcount, sizes_dev, 80, (10., 10., 5., 5.), 0.05, 0.5, 100, status=st)))
    print("whole step        %.3f ms" % timeit(lambda: model.inference_batched(batch)))
    lv, rois = K.assign_levels_rois(pboxes, 2, 5)
    print("levels hist", torch.bincount(lv.long(), minlength=4).tolist())
