"""Footprint statistics of the bench batch's 8 x 1000 proposals as ROIAlign sees them: level histogram, RoI size in feature pixels,
sampling grid, the window one (RoI, output row) workgroup stages, staged fraction, bytes fetched per scheme."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
lv = levels.cpu().numpy().astype(int); r = rois.cpu().numpy().astype(np.float64)
HW = [(f.shape[1], f.shape[2]) for f in flist]
sc = np.array(P.scales)[lv]
x1, y1, x2, y2 = [r[:, i + 1] * sc - 0.5 for i in range(4)]
w, h = x2 - x1, y2 - y1
gh, gw = np.ceil(h / 7).clip(1), np.ceil(w / 7).clip(1)
print("levels", np.bincount(lv, minlength=4), " RoI w feature px: median %.1f p90 %.1f max %.1f ; h: median %.1f p90 %.1f max %.1f" % (np.median(w), np.percentile(w, 90), w.max(), np.median(h), np.percentile(h, 90), h.max()))
print("grid gh x gw histogram (gh*gw):", np.bincount((gh * gw).astype(int))[:20], " mean samples/bin %.2f" % (gh * gw).mean())
H = np.array([HW[l][0] for l in lv]); W = np.array([HW[l][1] for l in lv])
ncols = np.minimum(np.floor(np.maximum(x1 + w, 0)) + 1, W - 1) - np.minimum(np.floor(np.maximum(x1, 0)), W - 1) + 1
tot_rows = 0; staged = 0; npix_staged = 0; npix_direct_taps = 0
for ph in range(7):
    yf, yl = y1 + ph * h / 7, y1 + (ph + 1) * h / 7
    nrows = np.minimum(np.floor(np.maximum(yl, 0)) + 1, H - 1) - np.minimum(np.floor(np.maximum(yf, 0)), H - 1) + 1
    npix = nrows * ncols
    st = npix <= 24
    staged += st.sum(); npix_staged += npix[st].sum(); npix_direct_taps += (7 * gh * gw * 4)[~st].sum()
    tot_rows += npix.sum()
foot = (np.minimum(np.floor(np.maximum(y1 + h, 0)) + 1, H - 1) - np.minimum(np.floor(np.maximum(y1, 0)), H - 1) + 1) * ncols
print("windows: %d of %d (RoI, row) workgroups staged; staged pixels %.1f MB, direct taps %.1f MB (L2/L1 reads), all windows %.1f MB; whole footprints once %.1f MB; output %.1f MB" % (
    staged, 7 * len(lv), npix_staged * 1024 / 1e6, npix_direct_taps * 1024 / 1e6, tot_rows * 1024 / 1e6, foot.sum() * 1024 / 1e6, len(lv) * 49 * 1024 / 1e6))
print("footprint pixels: median %d p90 %d max %d ; ncols median %d p90 %d" % (np.median(foot), np.percentile(foot, 90), foot.max(), np.median(ncols), np.percentile(ncols, 90)))
def timeit():
    run = lambda: K.roi_align_fpn_nhwc(flist, P.scales, rois, levels, 7, 7, P.sampling_ratio, P.aligned)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
print("roi_align alone: %.4f ms" % timeit())
