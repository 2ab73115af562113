"""Diagnose post-trunk chain differences: which proposal rows differ and why (near-tie margins)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import r50_state_dict
from test_gpu_chain import _chain_gpu, _r50
from lvc_amd.utils import synthetic as syn
from oracle import rcnn as orc

model = _r50(); sd = r50_state_dict(); spec = orc.RCNNSpec()
for seeds, hw in (((1, 2), (800, 1333)), ((3, 4), (320, 480))):
    inputs = [{"image": syn.synthetic_image(s, *hw), "height": hw[0], "width": hw[1]} for s in seeds]
    with torch.no_grad():
        imgs, sizes = orc.preprocess([b["image"] for b in inputs], spec.pixel_mean, spec.pixel_std, 32)
        feats = orc.fpn(sd, orc.resnet(sd, imgs, 50))
        ref, mid = orc.generalized_rcnn_inference(sd, spec, inputs, return_intermediates=True, feats=feats)
    pb, pl, pc, ob, osc, ocl, orow, cnt = _chain_gpu(model, feats, sizes, [hw] * len(inputs))
    for i in range(len(inputs)):
        rb, rl = mid["proposals"][i]
        n = min(pc[i], len(rb))
        d = (pb[i, :n] - rb[:n]).abs().max(dim=1)[0]
        bad = (d > 1e-3).nonzero().view(-1).tolist()
        print(hw, "image", i, "count", pc[i], len(rb), "rows differing", len(bad), bad[:20])
        for r in bad[:10]:
            # where is oracle row r in the gpu list?
            dd = (pb[i, :n] - rb[r]).abs().max(dim=1)[0]
            j = int(dd.argmin())
            print("   oracle row %d logit %.7f -> gpu row %d (d=%.1e) gpu logit there %.7f ; gpu row %d logit %.7f" % (r, float(rl[r]), j, float(dd[j]), float(pl[i, j]), r, float(pl[i, r])))
        r = ref[i]; m = cnt[i]
        print("   detections", m, len(r["scores"]), "classes equal", torch.equal(ocl[i, :m], r["pred_classes"]) if m == len(r["scores"]) else None,
              "rows equal", torch.equal(orow[i, :m], r["rows"]) if m == len(r["scores"]) else None)
        if m == len(r["scores"]):
            print("   worst score %.2e box %.2e" % (float((osc[i, :m] - r["scores"]).abs().max()), float((ob[i, :m] - r["pred_boxes"]).abs().max())))
            db = (ob[i, :m] - r["pred_boxes"]).abs().max(dim=1)[0]
            print("   det rows with box err > 1e-3:", (db > 1e-3).nonzero().view(-1).tolist()[:20])
